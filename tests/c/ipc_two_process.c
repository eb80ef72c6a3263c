/* Two processes, one C ABI, no Python, no torch, no RCCL: the multi-GPU plumbing of include/jxlgpu.h
 * ("multi-GPU: the stitched output without a collective") as a Rust / C host would drive it.
 *   owner  (parent): jxlgpu_device_alloc (two slots) -> jxlgpu_ipc_export -> the 64 handle bytes down a pipe;
 *                    renders a frame and formats it into slot 0 (JXLGPU_MEM_DEVICE destination)
 *   writer (child):  jxlgpu_ipc_open (a peer mapping; over xGMI when the processes sit on different GPUs) ->
 *                    renders the same frame -> jxlgpu_frame_format_output with dst = base + slot (its stores land in
 *                    the owner's allocation) -> jxlgpu_synchronize -> jxlgpu_ipc_close -> "done" up a pipe
 *   owner:           jxlgpu_device_download of the whole buffer; both slots must equal the frame formatted to host
 *                    memory by the ordinary path.
 * The processes are forked BEFORE either touches the GPU (a HIP runtime does not survive fork()).  Device of the
 * writer: JXLGPU_IPC_TEST_DEV (default 0: two processes on one GPU exercise the same export / open / peer-store code).
 * Exit codes: 0 ok, 3 no usable device, 1 anything else. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/types.h>
#include <sys/wait.h>
#include <unistd.h>

#include "jxlgpu.h"

enum { W = 64, H = 64 };

static int32_t g_coeff[3][H * W];
static float g_ones[64];
static int16_t g_lfq[3][64];
static uint8_t g_kind[64];
static int32_t g_hf_mul[64];
static float g_sigma[64];
static int32_t g_zero_tile[1];
static JxlGpuLfGroup g_group;

/* a 64x64 frame of 64 DCT8 varblocks with a few non-zero HF coefficients: enough structure for the formatted bytes to differ
 * from sample to sample */
static void make_desc(JxlGpuVardctDesc* d) {
    for (int i = 0; i < 64; ++i) {
        g_ones[i] = 1.0f;
        g_kind[i] = JXLGPU_DCT8;
        g_hf_mul[i] = 3 + i % 5;
        g_sigma[i] = 1.0f;
        g_lfq[0][i] = (int16_t)(40 + 3 * i);
        g_lfq[1][i] = (int16_t)(i % 7 - 3);
        g_lfq[2][i] = (int16_t)(17 + i);
    }
    for (int c = 0; c < 3; ++c)
        for (int i = 0; i < H * W; ++i) g_coeff[c][i] = (i % 8 == 1 && (i / W) % 8 == 0) ? (i % 5) - 2 : 0;
    memset(&g_group, 0, sizeof(g_group));
    g_group.width_px = W; g_group.height_px = H;
    for (int k = 0; k < 3; ++k) g_group.lf_quant[k] = g_lfq[k];
    g_group.extra_precision = 1;
    g_group.has_hf_meta = 1;
    g_group.block_kind = g_kind; g_group.hf_mul = g_hf_mul; g_group.epf_sigma = g_sigma;
    g_group.x_from_y = g_zero_tile; g_group.b_from_y = g_zero_tile;
    memset(d, 0, sizeof(*d));
    d->abi = JXLGPU_ABI_VERSION;
    d->width = W; d->height = H; d->group_dim = 256;
    d->lf_sample_type = JXLGPU_SAMPLE_I16;
    for (int c = 0; c < 3; ++c) d->coeff[c] = g_coeff[c];
    d->coeff_stride = W;
    d->coeff_format = JXLGPU_COEFF_DENSE; d->coeff_sample_type = JXLGPU_SAMPLE_I32;
    d->num_lf_groups = 1; d->lf_groups = &g_group;
    d->global_scale = 4096; d->quant_lf = 16;
    d->m_lf[0] = 1.0f / 32.0f; d->m_lf[1] = 1.0f / 4.0f; d->m_lf[2] = 1.0f / 2.0f;
    d->colour_factor = 84;
    d->x_factor_lf = 128; d->b_factor_lf = 128;
    d->x_qm_scale = 2; d->b_qm_scale = 2;
    d->quant_bias[0] = d->quant_bias[1] = d->quant_bias[2] = 0.5f;
    d->quant_bias_numerator = 0.145f;
    d->skip_adaptive_lf_smoothing = 1;
    for (int c = 0; c < 3; ++c) d->dequant[JXLGPU_DCT8][c] = g_ones;
    d->upsampling.factor = 1;
}

#define SLOT_BYTES ((size_t)W * H * 3 * 2)   /* u16 interleaved */

static int render_into(jxlgpu_ctx* ctx, void* dst, uint32_t mem, jxlgpu_frame** keep) {
    JxlGpuVardctDesc d;
    make_desc(&d);
    jxlgpu_frame* f = NULL;
    int rc = jxlgpu_vardct_upload(ctx, &d, &f);
    if (rc != JXLGPU_OK) { fprintf(stderr, "[%d] upload -> %d: %s\n", (int)getpid(), rc, jxlgpu_last_error(ctx)); return rc; }
    rc = jxlgpu_vardct_render(ctx, f, JXLGPU_STAGE_LF | JXLGPU_STAGE_TRANSFORM, NULL);
    if (rc != JXLGPU_OK) { fprintf(stderr, "[%d] render -> %d: %s\n", (int)getpid(), rc, jxlgpu_last_error(ctx)); return rc; }
    JxlGpuFormatDesc fmt;
    memset(&fmt, 0, sizeof(fmt));
    fmt.sample_format = JXLGPU_FMT_U16;
    fmt.orientation = 1;
    uint32_t ow = 0, oh = 0;
    rc = jxlgpu_frame_format_output(ctx, f, &fmt, dst, mem, &ow, &oh);
    if (rc != JXLGPU_OK) { fprintf(stderr, "[%d] format_output -> %d: %s\n", (int)getpid(), rc, jxlgpu_last_error(ctx)); return rc; }
    if (ow != W || oh != H) { fprintf(stderr, "format_output size %ux%u\n", ow, oh); return -1; }
    *keep = f;
    return JXLGPU_OK;
}

static int writer(int rd, int wr) {
    uint8_t handle[JXLGPU_IPC_HANDLE_BYTES];
    if (read(rd, handle, sizeof(handle)) != (ssize_t)sizeof(handle)) return 1;   /* the owner gave up */
    const char* dev = getenv("JXLGPU_IPC_TEST_DEV");
    jxlgpu_ctx* ctx = NULL;
    int rc = jxlgpu_create(dev ? atoi(dev) : 0, &ctx);
    if (rc != JXLGPU_OK) { fprintf(stderr, "writer: jxlgpu_create -> %d\n", rc); return rc == JXLGPU_ERR_DEVICE ? 3 : 1; }
    void* base = NULL;
    rc = jxlgpu_ipc_open(ctx, handle, &base);
    if (rc != JXLGPU_OK) { fprintf(stderr, "writer: ipc_open -> %d: %s\n", rc, jxlgpu_last_error(ctx)); jxlgpu_destroy(ctx); return 1; }
    jxlgpu_frame* f = NULL;
    rc = render_into(ctx, (char*)base + SLOT_BYTES, JXLGPU_MEM_DEVICE, &f);
    if (rc == JXLGPU_OK) rc = jxlgpu_synchronize(ctx);
    if (f) jxlgpu_frame_free(ctx, f);
    int rc2 = jxlgpu_ipc_close(ctx, base);
    jxlgpu_destroy(ctx);
    if (rc != JXLGPU_OK || rc2 != JXLGPU_OK) { fprintf(stderr, "writer: rc %d / ipc_close %d\n", rc, rc2); return 1; }
    const char ok = 'k';
    return write(wr, &ok, 1) == 1 ? 0 : 1;
}

int main(void) {
    if (jxlgpu_abi_version() != JXLGPU_ABI_VERSION) { fprintf(stderr, "ABI mismatch\n"); return 1; }
    int p2c[2], c2p[2];
    if (pipe(p2c) || pipe(c2p)) { perror("pipe"); return 1; }
    const pid_t pid = fork();   /* before any GPU call in either process */
    if (pid < 0) { perror("fork"); return 1; }
    if (pid == 0) {
        close(p2c[1]); close(c2p[0]);
        _exit(writer(p2c[0], c2p[1]));
    }
    close(p2c[0]); close(c2p[1]);
    int result = 1, status = 0;
    jxlgpu_ctx* ctx = NULL;
    void* buf = NULL;
    jxlgpu_frame* f = NULL;
    int rc = jxlgpu_create(0, &ctx);
    if (rc != JXLGPU_OK) {
        fprintf(stderr, "owner: jxlgpu_create -> %d (no CPU fallback)\n", rc);
        close(p2c[1]);            /* the writer's read() returns 0 and it leaves */
        waitpid(pid, &status, 0);
        return rc == JXLGPU_ERR_DEVICE ? 3 : 1;
    }
    uint8_t handle[JXLGPU_IPC_HANDLE_BYTES];
    static uint8_t stitched[2 * SLOT_BYTES], expect[SLOT_BYTES];
    do {
        if ((rc = jxlgpu_device_alloc(ctx, 2 * SLOT_BYTES, &buf)) != JXLGPU_OK) { fprintf(stderr, "device_alloc -> %d\n", rc); break; }
        if ((rc = jxlgpu_ipc_export(ctx, buf, handle)) != JXLGPU_OK) { fprintf(stderr, "ipc_export -> %d: %s\n", rc, jxlgpu_last_error(ctx)); break; }
        if (write(p2c[1], handle, sizeof(handle)) != (ssize_t)sizeof(handle)) { perror("write handle"); break; }
        if (render_into(ctx, buf, JXLGPU_MEM_DEVICE, &f) != JXLGPU_OK) break;          /* slot 0: the owner's own frame */
        char ok = 0;
        if (read(c2p[0], &ok, 1) != 1 || ok != 'k') { fprintf(stderr, "owner: the writer did not finish\n"); break; }
        if ((rc = jxlgpu_device_download(ctx, buf, stitched, sizeof(stitched))) != JXLGPU_OK) { fprintf(stderr, "device_download -> %d\n", rc); break; }
        JxlGpuFormatDesc fmt;
        memset(&fmt, 0, sizeof(fmt));
        fmt.sample_format = JXLGPU_FMT_U16; fmt.orientation = 1;
        if ((rc = jxlgpu_frame_format_output(ctx, f, &fmt, expect, JXLGPU_MEM_HOST, NULL, NULL)) != JXLGPU_OK) { fprintf(stderr, "format to host -> %d\n", rc); break; }
        int nonzero = 0;
        for (size_t i = 0; i < SLOT_BYTES; ++i) nonzero |= expect[i];
        if (!nonzero) { fprintf(stderr, "the formatted frame is all zero: the test would prove nothing\n"); break; }
        if (memcmp(stitched, expect, SLOT_BYTES)) { fprintf(stderr, "slot 0 (owner's own stores) differs\n"); break; }
        if (memcmp(stitched + SLOT_BYTES, expect, SLOT_BYTES)) { fprintf(stderr, "slot 1 (the writer's stores through the IPC mapping) differs\n"); break; }
        result = 0;
    } while (0);
    close(p2c[1]);
    waitpid(pid, &status, 0);
    if (f) jxlgpu_frame_free(ctx, f);
    if (buf) jxlgpu_device_free(ctx, buf);
    jxlgpu_destroy(ctx);
    if (result == 0 && (!WIFEXITED(status) || WEXITSTATUS(status) != 0)) {
        fprintf(stderr, "writer exit status %d\n", WIFEXITED(status) ? WEXITSTATUS(status) : -1);
        result = 1;
    }
    if (result == 0) printf("ok\n");
    return result;
}
