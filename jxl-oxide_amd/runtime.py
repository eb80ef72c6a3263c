"""Thin Python host over the C ABI (ctypes): plumbing for tests, smoke() and bench.py.
Every call goes through libjxlgpu.so; there is no CPU fallback here."""
import ctypes as C

import numpy as np

from . import abi


def canary_path():
    import os
    return os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "_bin", "canary")


def gpu_canary(attempts=3, timeout=120, quiet=False):
    """Box or library?  Runs tests/c/canary.hip (built by __graft_entry__.build() into tools/_bin/canary): a pure-HIP
    program — no libjxlgpu.so, no torch — that uses the runtime the way the library does (small exact-size
    hipMallocs, non-blocking streams, blocking / async / 2-D copies, pinned staging, > 64 KiB dynamic LDS) with
    trivially in-bounds kernels.  If THAT dies with "Memory access fault by GPU node", the box is at fault: nothing
    of ours was loaded.  Up to `attempts` fresh processes; every attempt is printed ("CANARY attempt k: ...") so
    that the driver's log says which it was.  Returns "ok" (first attempt clean), "ok-after-fault" (an earlier
    attempt died, a later one ran clean: the first-process fault of some fresh boxes), "fault" (every attempt
    died), "nodevice" or "missing" (binary not built)."""
    import os
    import subprocess
    if os.environ.get("JXLGPU_NO_CANARY"):  # profiling runs: rocprofv3 would follow the child process
        return "skipped"
    exe = canary_path()
    if not os.path.exists(exe):
        # normally built by __graft_entry__.build(); the GPU boxes carry the same toolchain
        src = os.path.join(os.path.dirname(os.path.dirname(exe)), "..", "tests", "c", "canary.hip")
        try:
            os.makedirs(os.path.dirname(exe), exist_ok=True)
            subprocess.run(["/opt/rocm/bin/hipcc", "-O2", "--offload-arch=gfx950", os.path.normpath(src), "-o", exe],
                           timeout=300, capture_output=True, check=True)
        except Exception:  # noqa: BLE001
            pass
    if not os.path.exists(exe):
        if not quiet:
            print("CANARY missing: tools/_bin/canary not built (run __graft_entry__.build())", flush=True)
        return "missing"
    seen_fault = False
    for k in range(attempts):
        try:
            r = subprocess.run([exe], timeout=timeout, capture_output=True, text=True)
            rc, out, err = r.returncode, r.stdout.strip(), r.stderr.strip()
        except subprocess.TimeoutExpired:
            rc, out, err = -999, "", "timeout"
        if not quiet:
            print(f"CANARY attempt {k + 1}: rc={rc} {out} {err[:300]}".rstrip(), flush=True)
        if rc == 0 and out.endswith("CANARY ok"):
            return "ok-after-fault" if seen_fault else "ok"
        if rc == 3:
            return "nodevice"
        seen_fault = True
    return "fault"


class JxlGpuError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"jxlgpu error {code}: {msg}")
        self.code = code


class Frame:
    def __init__(self, ctx, handle):
        self.ctx, self.handle = ctx, handle

    def free(self):
        if self.handle:
            self.ctx.lib.jxlgpu_frame_free(self.ctx.handle, self.handle)
            self.handle = None

    def out_size(self, stages):
        w, h = C.c_uint32(), C.c_uint32()
        self.ctx._check(self.ctx.lib.jxlgpu_frame_out_size(self.handle, stages, C.byref(w), C.byref(h)))
        return w.value, h.value

    def result_size(self):
        """(width, height) of the frame's last render (a region render: the region's)."""
        w, h = C.c_uint32(), C.c_uint32()
        self.ctx._check(self.ctx.lib.jxlgpu_frame_result_size(self.handle, C.byref(w), C.byref(h)))
        return w.value, h.value

    def algorithmic_bytes(self, stages):
        return int(self.ctx.lib.jxlgpu_frame_algorithmic_bytes(self.handle, stages))

    def result_plane_ptr(self, c):
        return C.cast(self.ctx.lib.jxlgpu_frame_result_plane(self.handle, c), C.c_void_p).value


class Context:
    def __init__(self, device=0):
        self.lib = abi.load_library()
        self.device = device
        if self.lib.jxlgpu_abi_version() != abi.ABI_VERSION:
            raise RuntimeError("libjxlgpu.so ABI version does not match abi.py")
        h = C.c_void_p()
        rc = self.lib.jxlgpu_create(device, C.byref(h))
        if rc != abi.OK:
            raise JxlGpuError(rc, "jxlgpu_create failed (no MI355X visible?)")
        self.handle = h

    def set_trace(self, callback):
        """jxlgpu_set_trace: `callback(span: str, begin: bool)` on the calling thread around every launch group, with the reference's
        span name (None switches it off).  The ctypes thunk is kept alive on the context."""
        if callback is None:
            self._trace = None
            self._check(self.lib.jxlgpu_set_trace(self.handle, None, None))
            return
        self._trace = abi.TRACE_FN(lambda user, span, begin: callback(span.decode(), bool(begin)))
        self._check(self.lib.jxlgpu_set_trace(self.handle, C.cast(self._trace, C.c_void_p), None))

    def close(self):
        if self.handle:
            self.lib.jxlgpu_destroy(self.handle)
            self.handle = None

    def _check(self, rc):
        if rc != abi.OK:
            raise JxlGpuError(rc, self.lib.jxlgpu_last_error(self.handle).decode())

    def stream(self):
        return self.lib.jxlgpu_stream(self.handle)

    def synchronize(self):
        self._check(self.lib.jxlgpu_synchronize(self.handle))

    def profile_select(self, group):
        self._check(self.lib.jxlgpu_profile_select(self.handle, group))

    def profile_read(self):
        ms, n = C.c_double(), C.c_uint64()
        self._check(self.lib.jxlgpu_profile_read(self.handle, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    # ---- VarDCT
    def vardct_upload(self, desc):
        fh = C.c_void_p()
        self._check(self.lib.jxlgpu_vardct_upload(self.handle, C.byref(desc), C.byref(fh)))
        return Frame(self, fh)

    def vardct_render(self, frame, stages, to_host=True):
        """Runs the stages; returns planes[3][h, w] (numpy) or None when to_host is False."""
        if not to_host:
            self._check(self.lib.jxlgpu_vardct_render(self.handle, frame.handle, stages, None))
            return None
        w, h = frame.out_size(stages)
        out = np.zeros((3, h, w), dtype=np.float32)
        o = abi.Out()
        for c in range(3):
            o.planes[c] = out[c].ctypes.data_as(abi.f32p)
        o.stride = w
        o.mem = abi.MEM_HOST
        self._check(self.lib.jxlgpu_vardct_render(self.handle, frame.handle, stages, C.byref(o)))
        return out

    def _render_region(self, fn, frame, stages, region, to_host=True):
        left, top, width, height = region
        if not to_host:  # the result stays on the device (jxlgpu_frame_result_plane / jxlgpu_frame_format_output)
            r = abi.Region(left, top, width, height)
            self._check(fn(self.handle, frame.handle, stages, C.byref(r), None))
            return None
        out = np.zeros((3, height, width), dtype=np.float32)
        o = abi.Out()
        for c in range(3):
            o.planes[c] = out[c].ctypes.data_as(abi.f32p)
        o.stride = width
        o.mem = abi.MEM_HOST
        r = abi.Region(left, top, width, height)
        self._check(fn(self.handle, frame.handle, stages, C.byref(r), C.byref(o)))
        # the library clips the region to the frame and writes the intersection at the origin of `out`
        rw, rh = frame.result_size()
        return out[:, :rh, :rw]

    def vardct_render_region(self, frame, stages, region, to_host=True):
        """`region` = (left, top, width, height) of the output, inside the frame -> planes[3][height, width]."""
        return self._render_region(self.lib.jxlgpu_vardct_render_region, frame, stages, region, to_host)

    def modular_render_region(self, frame, stages, region, to_host=True):
        return self._render_region(self.lib.jxlgpu_modular_render_region, frame, stages, region, to_host)

    def vardct_render_batch(self, frames, stages):
        """One launch per stage for all `frames` (asynchronous; results stay on the device)."""
        arr = (C.c_void_p * len(frames))(*[f.handle for f in frames])
        self._check(self.lib.jxlgpu_vardct_render_batch(self.handle, arr, len(frames), stages))

    def download_result(self, frame, stages=None):
        """Planar f32 result of the frame's last render, at the size that render produced (`stages` is
        accepted for older callers and ignored)."""
        cw, ch = C.c_uint32(), C.c_uint32()
        self._check(self.lib.jxlgpu_frame_result_size(frame.handle, C.byref(cw), C.byref(ch)))
        w, h = cw.value, ch.value
        out = np.zeros((3, h, w), dtype=np.float32)
        o = abi.Out()
        for c in range(3):
            o.planes[c] = out[c].ctypes.data_as(abi.f32p)
        o.stride = w
        o.mem = abi.MEM_HOST
        self._check(self.lib.jxlgpu_frame_download_result(self.handle, frame.handle, C.byref(o)))
        return out

    def vardct_render_host(self, desc, stages, out_w, out_h, out=None):
        """One-shot host-to-host call.  `out`: optional preallocated (3, out_h, out_w) float32 array
        (a decoder reuses its frame buffers; a fresh np.zeros pays a page fault per 4 KB)."""
        if out is None:
            out = np.zeros((3, out_h, out_w), dtype=np.float32)
        o = abi.Out()
        for c in range(3):
            o.planes[c] = out[c].ctypes.data_as(abi.f32p)
        o.stride = out_w
        o.mem = abi.MEM_HOST
        self._check(self.lib.jxlgpu_vardct_render_host(self.handle, C.byref(desc), stages, C.byref(o)))
        return out

    def blend_rects(self, base_ptr, base_stride, base_w, base_h, new_ptr, new_stride, new_w, new_h, rects):
        """blend_single on device planes (raw device pointers); `rects`: list of abi.BlendRect."""
        arr = (abi.BlendRect * len(rects))(*rects)
        self._check(self.lib.jxlgpu_blend_rects(self.handle, base_ptr, base_stride, base_w, base_h, new_ptr,
                                                new_stride, new_w, new_h, arr, len(rects)))

    def render_extra(self, frame, index, ec, to_host=True):
        """Extra channel `index` of the frame: int -> float with its own bit depth + non-separable upsampling; the plane
        stays on the device with the frame (format_output interleaves it).  Returns it as (h, w) f32 if to_host."""
        w, h = ec.width << ec.upsampling_log2, ec.height << ec.upsampling_log2
        if not to_host:
            self._check(self.lib.jxlgpu_frame_render_extra(self.handle, frame.handle, index, C.byref(ec), None, 0, abi.MEM_HOST))
            return None
        out = np.zeros((h, w), dtype=np.float32)
        self._check(self.lib.jxlgpu_frame_render_extra(self.handle, frame.handle, index, C.byref(ec), out.ctypes.data, w, abi.MEM_HOST))
        return out

    def format_output(self, frame, sample_format, orientation=1, extra=()):
        """Interleaved, oriented f32/u16/u8 image of the last render — a whole frame or a region —
        formatted on the device; `extra`: indices of rendered extra channels appended to every pixel (RGBA)."""
        w, h = frame.result_size()
        ow, oh = (w, h) if orientation <= 4 else (h, w)
        dt = {abi.FMT_F32: np.float32, abi.FMT_U16: np.uint16, abi.FMT_U8: np.uint8}[sample_format]
        out = np.zeros((oh, ow, 3 + len(extra)), dtype=dt)
        fmt = abi.FormatDesc(sample_format, orientation)
        fmt.num_extra = len(extra)
        for i, e in enumerate(extra):
            fmt.extra[i] = e
        rw, rh = C.c_uint32(), C.c_uint32()
        self._check(self.lib.jxlgpu_frame_format_output(self.handle, frame.handle, C.byref(fmt), out.ctypes.data,
                                                        abi.MEM_HOST, C.byref(rw), C.byref(rh)))
        assert (rw.value, rh.value) == (ow, oh)
        return out

    def host_alloc(self, shape, dtype):
        """Pinned host memory (jxlgpu_host_alloc) as a numpy array; free it with host_free(arr)."""
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p = C.c_void_p()
        self._check(self.lib.jxlgpu_host_alloc(self.handle, n, C.byref(p)))
        buf = (C.c_char * n).from_address(p.value)
        arr = np.frombuffer(buf, dtype=dtype).reshape(shape)
        self._pinned = getattr(self, "_pinned", {})
        self._pinned[arr.ctypes.data] = p.value
        return arr

    def host_free(self, arr):
        p = getattr(self, "_pinned", {}).pop(arr.ctypes.data, None)
        if p:
            self.lib.jxlgpu_host_free(self.handle, p)

    def format_output_async(self, frame, sample_format, out_pinned, orientation=1):
        """format_output into pinned memory (host_alloc): returns once the work is queued; frame_wait(frame)
        (or synchronize) tells when `out_pinned` is complete."""
        fmt = abi.FormatDesc(sample_format, orientation)
        rw, rh = C.c_uint32(), C.c_uint32()
        self._check(self.lib.jxlgpu_frame_format_output(self.handle, frame.handle, C.byref(fmt), out_pinned.ctypes.data,
                                                        abi.MEM_HOST_PINNED, C.byref(rw), C.byref(rh)))
        return rw.value, rh.value

    def set_memory_limit(self, nbytes):
        self._check(self.lib.jxlgpu_set_memory_limit(self.handle, int(nbytes)))

    def memory_usage(self):
        live, pooled = C.c_uint64(), C.c_uint64()
        self._check(self.lib.jxlgpu_memory_usage(self.handle, C.byref(live), C.byref(pooled)))
        return live.value, pooled.value

    # ---- multi-GPU plumbing (no torch involved)
    def device_alloc(self, nbytes):
        p = C.c_void_p()
        self._check(self.lib.jxlgpu_device_alloc(self.handle, nbytes, C.byref(p)))
        return p.value

    def device_free(self, ptr):
        self.lib.jxlgpu_device_free(self.handle, ptr)

    def ipc_export(self, ptr):
        h = (C.c_uint8 * abi.IPC_HANDLE_BYTES)()
        self._check(self.lib.jxlgpu_ipc_export(self.handle, ptr, h))
        return bytes(h)

    def ipc_open(self, handle_bytes):
        h = (C.c_uint8 * abi.IPC_HANDLE_BYTES)(*handle_bytes)
        p = C.c_void_p()
        self._check(self.lib.jxlgpu_ipc_open(self.handle, h, C.byref(p)))
        return p.value

    def ipc_close(self, ptr):
        self._check(self.lib.jxlgpu_ipc_close(self.handle, ptr))

    def device_download(self, ptr, shape, dtype):
        out = np.zeros(shape, dtype=dtype)
        self._check(self.lib.jxlgpu_device_download(self.handle, ptr, out.ctypes.data, out.nbytes))
        return out

    def device_upload(self, ptr, arr):
        arr = np.ascontiguousarray(arr)
        self._check(self.lib.jxlgpu_device_upload(self.handle, ptr, arr.ctypes.data, arr.nbytes))

    def format_output_to(self, frame, sample_format, dev_ptr, orientation=1):
        """format_output with a device destination (own memory or a peer mapping from ipc_open); asynchronous."""
        fmt = abi.FormatDesc(sample_format, orientation)
        rw, rh = C.c_uint32(), C.c_uint32()
        self._check(self.lib.jxlgpu_frame_format_output(self.handle, frame.handle, C.byref(fmt), dev_ptr, abi.MEM_DEVICE,
                                                        C.byref(rw), C.byref(rh)))
        return rw.value, rh.value

    def frame_wait(self, frame):
        self._check(self.lib.jxlgpu_frame_wait(self.handle, frame.handle))

    def upload_split(self):
        """Host-time split of the last vardct_upload (ms): build, -, alloc + enqueue, whole call, H2D on the device."""
        ms = (C.c_double * 5)()
        self._check(self.lib.jxlgpu_upload_split(self.handle, ms))
        return list(ms)

    def selftest_libm(self, which, x, y=1.0):
        """The device's logf (which = 0) / powf(x, y) (which = 1) restatements (csrc/libm_f32.h) on the f32 array `x`."""
        x = np.ascontiguousarray(x, dtype=np.float32)
        out = np.empty_like(x)
        self._check(self.lib.jxlgpu_selftest_libm(self.handle, which, x.ctypes.data_as(abi.f32p), x.size, y,
                                                  out.ctypes.data_as(abi.f32p)))
        return out

    def download_lf(self, frame, w8, h8):
        lf = np.zeros((3, h8, w8), dtype=np.float32)
        arr = (abi.f32p * 3)(*[lf[c].ctypes.data_as(abi.f32p) for c in range(3)])
        self._check(self.lib.jxlgpu_frame_download_lf(self.handle, frame.handle, arr))
        return lf

    # ---- Modular
    def modular_upload(self, desc):
        fh = C.c_void_p()
        self._check(self.lib.jxlgpu_modular_upload(self.handle, C.byref(desc), C.byref(fh)))
        return Frame(self, fh)

    def modular_inverse(self, frame, shapes, dtype, to_host=True):
        """Inverse transforms only; returns the reconstructed integer planes (bit-exact contract)."""
        if not to_host:
            self._check(self.lib.jxlgpu_modular_inverse(self.handle, frame.handle, None))
            return None
        outs = [np.zeros(s, dtype=dtype) for s in shapes]
        arr = (C.c_void_p * len(outs))(*[o.ctypes.data for o in outs])
        self._check(self.lib.jxlgpu_modular_inverse(self.handle, frame.handle, arr))
        return outs

    def modular_render(self, frame, stages, to_host=True):
        if not to_host:
            self._check(self.lib.jxlgpu_modular_render(self.handle, frame.handle, stages, None))
            return None
        w, h = frame.out_size(stages)
        out = np.zeros((3, h, w), dtype=np.float32)
        o = abi.Out()
        for c in range(3):
            o.planes[c] = out[c].ctypes.data_as(abi.f32p)
        o.stride = w
        o.mem = abi.MEM_HOST
        self._check(self.lib.jxlgpu_modular_render(self.handle, frame.handle, stages, C.byref(o)))
        return out
