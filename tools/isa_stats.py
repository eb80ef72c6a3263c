#!/usr/bin/env python3
"""Static instruction mix per kernel from `hipcc --cuda-device-only -S` output.
    hipcc <flags> --cuda-device-only -S file.hip -o /tmp/x.s && python tools/isa_stats.py /tmp/x.s"""
import collections, re, sys
cur = None
stats = collections.OrderedDict()
for line in open(sys.argv[1]):
    m = re.match(r'^(_Z\S+):', line)
    if m:
        cur = m.group(1); stats[cur] = collections.Counter(); continue
    if line.startswith('.Lfunc_end'):
        cur = None; continue
    m = re.match(r'^\s+([a-z_0-9]+)\s', line)
    if cur and m:
        op = m.group(1); st = stats[cur]; st['total'] += 1
        if op.startswith('v_accvgpr'): st['accvgpr'] += 1
        elif op.startswith('v_'): st['valu'] += 1
        elif op.startswith('ds_'): st['lds'] += 1
        elif op.startswith(('global_', 'buffer_', 'flat_')): st['vmem'] += 1
        elif op.startswith('scratch_'): st['scratch'] += 1
        elif op.startswith(('s_cbranch', 's_branch')): st['branch'] += 1
        elif op.startswith('s_waitcnt'): st['wait'] += 1
        elif op.startswith('s_'): st['salu'] += 1
for k, v in stats.items():
    print(k[:90], dict(v))
