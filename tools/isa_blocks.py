import re,sys
lines=open(sys.argv[1]).read().split('\n')
name=sys.argv[2]
start=[i for i,l in enumerate(lines) if l.startswith(name) and ': ' in l+' ' and l.split(':')[0]==name][0]
end=[i for i in range(start,len(lines)) if lines[i].startswith('.Lfunc_end')][0]
blk='entry'; blocks={}; order=[]
for l in lines[start+1:end]:
    m=re.match(r'^(\.LBB\d+_\d+):',l)
    if m: blk=m.group(1)
    if blk not in blocks: blocks[blk]={'valu':0,'pk':0,'div':0,'dpp':0,'mov':0,'br':[]}; order.append(blk)
    m=re.match(r'^\s+([a-z_0-9]+)\s*(.*)',l)
    if not m: continue
    op=m.group(1); b=blocks[blk]
    if op.startswith('v_'): b['valu']+=1
    if op.startswith('v_pk_'): b['pk']+=1
    if op.startswith('v_div_'): b['div']+=1
    if op.startswith('v_mov') or op.startswith('v_accvgpr'): b['mov']+=1
    if 'dpp' in op or 'row_' in l or 'wave_sh' in l: b['dpp']+=1
    if op.startswith('s_cbranch') or op.startswith('s_branch'):
        b['br'].append(m.group(2).strip()); blk=blk+"'"
for k in order:
    if blocks[k]['valu']>2: print(k, blocks[k])
