"""Host-side mirror of the reference's filter interface over the C ABI of libmvtools_amd.so.

    core.mv.Super(clip, ...)            -> Super(width, height, bits, ...)       .build(frames)
    core.mv.Analyse(super, ...)         -> Analyse(super, num_frames, ...)       .run(jobs)
    core.mv.Degrain1..6(clip, super, mvbw, mvfw, ...) -> Degrain(radius, super, analysis_data, ...) .run(jobs)
    core.mv.Compensate(clip, super, vectors, ...)     -> Compensate(super, analysis_data, ...)      .run(jobs)

Argument names, defaults and error strings are the reference's (MVSuper.c:279-291, MVAnalyse.c:639-671,
MVDegrains.cpp:813-932, MVCompensate.c:579-592); they are resolved inside the library, not here.
PyTorch is only plumbing: device memory (uint8 tensors, one per plane, row stride = pitch) and the HIP stream.
All pixel work happens in the hand-written HIP kernels; there is no CPU fallback -- a missing library or a missing
GPU raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBPATH = os.environ.get("MVX_LIB") or os.path.join(os.path.dirname(_HERE), "libmvtools_amd.so")  # MVX_LIB: developer override (A/B builds)
UNSET = -2147483648
ERRLEN = 256


class MvtoolsError(Exception):
    pass


class AnalysisData(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "nMagicKey", "nVersion", "nBlkSizeX", "nBlkSizeY", "nPel", "nLvCount", "nDeltaFrame", "isBackward", "nCPUFlags",
        "nMotionFlags", "nWidth", "nHeight", "nOverlapX", "nOverlapY", "nBlkX", "nBlkY", "bitsPerSample", "yRatioUV",
        "xRatioUV", "nHPadding", "nVPadding")]


class SuperArgs(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("width", "height", "bits", "subsampling_w", "subsampling_h", "gray", "hpad", "vpad",
                                         "pel", "levels", "chroma", "sharp", "rfilter")]


class SuperInfo(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("width", "height", "bits", "xRatioUV", "yRatioUV", "gray", "hpad", "vpad", "pel",
                                         "levels", "chroma", "sharp", "rfilter", "modeYUV", "super_width", "super_height",
                                         "num_planes")] + [("plane_width", C.c_int32 * 3), ("plane_height", C.c_int32 * 3)]


ANALYSE_ARGS = ("blksize", "blksizev", "levels", "search", "searchparam", "pelsearch", "isb", "lambda_", "chroma", "delta",
                "truemotion", "lsad", "plevel", "global_", "pnew", "pzero", "pglobal", "overlap", "overlapv", "divide", "badsad",
                "badrange", "opt", "meander", "trymany", "fields", "tff", "search_coarse", "dct")


class AnalyseArgs(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ANALYSE_ARGS]


class AnalyseJob(C.Structure):
    _fields_ = [("src", C.c_void_p * 3), ("ref", C.c_void_p * 3), ("blob", C.c_void_p), ("field_shift", C.c_int32), ("reserved", C.c_int32)]


class DegrainArgs(C.Structure):
    _fields_ = [("radius", C.c_int32), ("thsad", C.c_int64), ("thsadc", C.c_int64), ("plane", C.c_int32), ("limit", C.c_int32),
                ("limitc", C.c_int32), ("thscd1", C.c_int64), ("thscd2", C.c_int32)]


class DegrainJob(C.Structure):
    _fields_ = [("src", C.c_void_p * 3), ("refs", (C.c_void_p * 3) * 12), ("blobs", C.c_void_p * 12), ("dst", C.c_void_p * 3)]


class CompensateArgs(C.Structure):
    _fields_ = [("scbehavior", C.c_int32), ("thsad", C.c_int64), ("time", C.c_double), ("thscd1", C.c_int64), ("thscd2", C.c_int32),
                ("fields", C.c_int32)]


class CompensateJob(C.Structure):
    _fields_ = [("src_super", C.c_void_p * 3), ("ref_super", C.c_void_p * 3), ("blob", C.c_void_p), ("dst", C.c_void_p * 3),
                ("field_shift", C.c_int32), ("reserved", C.c_int32)]


RECALC_ARGS = ("thsad", "smooth", "blksize", "blksizev", "search", "searchparam", "lambda_", "chroma", "truemotion", "pnew", "overlap", "overlapv", "divide",
               "meander", "fields", "dct")


class RecalculateArgs(C.Structure):
    _fields_ = [(n, C.c_int64) for n in RECALC_ARGS]


class RecalculateJob(C.Structure):
    _fields_ = [("src", C.c_void_p * 3), ("ref", C.c_void_p * 3), ("old_blob", C.c_void_p), ("blob", C.c_void_p)]


class BlockFPSArgs(C.Structure):
    _fields_ = [("num", C.c_int64), ("den", C.c_int64), ("mode", C.c_int32), ("ml", C.c_double), ("blend", C.c_int32), ("thscd1", C.c_int64), ("thscd2", C.c_int32)]


class BlockFPSInfo(C.Structure):
    _fields_ = [("num_frames", C.c_int32), ("fps_num", C.c_int64), ("fps_den", C.c_int64)]


class BlockFPSJob(C.Structure):
    _fields_ = [("time256", C.c_int32), ("reserved", C.c_int32), ("src_super", C.c_void_p * 3), ("ref_super", C.c_void_p * 3), ("blob_fw", C.c_void_p),
                ("blob_bw", C.c_void_p), ("clip_left", C.c_void_p * 3), ("clip_right", C.c_void_p * 3), ("dst", C.c_void_p * 3)]


_lib = None


def lib():
    """Loads libmvtools_amd.so (built by vapoursynth-mvtools_amd/build.py).  Fails loudly when it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIBPATH):
            raise MvtoolsError("libmvtools_amd.so not built (%s): run `python vapoursynth-mvtools_amd/build.py`; "
                               "there is no CPU fallback" % _LIBPATH)
        # One HIP runtime per process: PyTorch bundles its own libamdhip64.so.7.  Importing torch first makes the
        # dynamic loader bind our DT_NEEDED libamdhip64.so.7 to that already-loaded copy instead of pulling a second
        # runtime from /opt/rocm (two HSA runtimes in one process cannot both open the device).
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(_LIBPATH)
        P = C.POINTER
        L.mvx_last_error.restype = C.c_char_p
        L.mvx_version.restype = C.c_char_p
        L.mvx_super_create.argtypes = [P(SuperArgs), P(C.c_void_p), C.c_char_p]
        L.mvx_super_destroy.argtypes = [C.c_void_p]
        L.mvx_super_get_info.argtypes = [C.c_void_p, P(SuperInfo)]
        L.mvx_super_frames.argtypes = [C.c_void_p, C.c_int, P(C.c_void_p), P(C.c_ssize_t), P(C.c_void_p), P(C.c_ssize_t), C.c_void_p]
        L.mvx_super_pelclip_mode.argtypes = [C.c_void_p, C.c_int, C.c_int, P(C.c_int32), C.c_char_p]
        L.mvx_super_frames_pelclip.argtypes = [C.c_void_p, C.c_int, P(C.c_void_p), P(C.c_ssize_t), P(C.c_void_p), P(C.c_ssize_t), C.c_int, P(C.c_void_p),
                                               P(C.c_ssize_t), C.c_void_p]
        L.mvx_super_shadow_copies.argtypes = [C.c_void_p]
        L.mvx_super_shadow_bytes.argtypes = [C.c_void_p, P(C.c_ssize_t), P(C.c_size_t)]
        L.mvx_super_shadow_bytes.restype = None
        L.mvx_super_shadow_frames.argtypes = [C.c_void_p, C.c_int, P(C.c_void_p), P(C.c_ssize_t), P(C.c_ssize_t), C.c_void_p]
        L.mvx_analyse_set_ref_shadow.argtypes = [C.c_void_p, P(C.c_ssize_t)]
        L.mvx_degrain_set_ref_shadow.argtypes = [C.c_void_p, P(C.c_ssize_t)]
        L.mvx_super_frames_shadow.argtypes = [C.c_void_p, C.c_int, P(C.c_void_p), P(C.c_ssize_t), P(C.c_void_p), P(C.c_ssize_t), P(C.c_ssize_t), C.c_void_p]
        L.mvx_debug_option.argtypes = [C.c_char_p, C.c_int]
        L.mvx_analyse_create.argtypes = [P(AnalyseArgs), C.c_void_p, C.c_int, P(C.c_ssize_t), P(C.c_void_p), C.c_char_p]
        L.mvx_analyse_destroy.argtypes = [C.c_void_p]
        L.mvx_analyse_get_data.argtypes = [C.c_void_p, P(AnalysisData)]
        L.mvx_analyse_blob_size.argtypes = [C.c_void_p]
        L.mvx_analyse_frames.argtypes = [C.c_void_p, C.c_int, P(AnalyseJob), C.c_void_p]
        L.mvx_degrain_create.argtypes = [P(DegrainArgs), P(AnalysisData), C.c_void_p, P(C.c_ssize_t), P(C.c_ssize_t), P(C.c_ssize_t),
                                         P(C.c_void_p), C.c_char_p]
        L.mvx_degrain_destroy.argtypes = [C.c_void_p]
        L.mvx_degrain_frames.argtypes = [C.c_void_p, C.c_int, P(DegrainJob), C.c_void_p]
        L.mvx_compensate_create.argtypes = [P(CompensateArgs), P(AnalysisData), C.c_void_p, P(C.c_ssize_t), P(C.c_ssize_t),
                                            P(C.c_void_p), C.c_char_p]
        L.mvx_compensate_destroy.argtypes = [C.c_void_p]
        L.mvx_compensate_frames.argtypes = [C.c_void_p, C.c_int, P(CompensateJob), C.c_void_p]
        L.mvx_finest_size.argtypes = [C.c_void_p, P(C.c_int32), P(C.c_int32)]
        L.mvx_finest_frames.argtypes = [C.c_void_p, C.c_int, P(C.c_void_p), P(C.c_ssize_t), P(C.c_void_p), P(C.c_ssize_t), C.c_void_p]
        L.mvx_scdetect.argtypes = [P(AnalysisData), C.c_int64, C.c_int32, C.c_int, P(C.c_void_p), P(C.c_int32), C.c_void_p, C.c_char_p]
        L.mvx_recalculate_create.argtypes = [P(RecalculateArgs), C.c_void_p, P(AnalysisData), P(C.c_ssize_t), P(C.c_void_p), C.c_char_p]
        L.mvx_recalculate_destroy.argtypes = [C.c_void_p]
        L.mvx_recalculate_get_data.argtypes = [C.c_void_p, P(AnalysisData)]
        L.mvx_recalculate_blob_size.argtypes = [C.c_void_p]
        L.mvx_recalculate_frames.argtypes = [C.c_void_p, C.c_int, P(RecalculateJob), C.c_void_p]
        L.mvx_blockfps_create.argtypes = [P(BlockFPSArgs), P(AnalysisData), P(AnalysisData), C.c_void_p, C.c_int, C.c_int64, C.c_int64, P(C.c_ssize_t),
                                          P(C.c_ssize_t), P(C.c_ssize_t), P(C.c_void_p), C.c_char_p]
        L.mvx_blockfps_destroy.argtypes = [C.c_void_p]
        L.mvx_blockfps_get_info.argtypes = [C.c_void_p, P(BlockFPSInfo)]
        L.mvx_blockfps_map.argtypes = [C.c_void_p, C.c_int, P(C.c_int), P(C.c_int), P(C.c_int)]
        L.mvx_blockfps_frames.argtypes = [C.c_void_p, C.c_int, P(BlockFPSJob), C.c_void_p]
        L.mvx_scale_thscd.argtypes = [P(C.c_int64), P(C.c_int32), P(AnalysisData)]
        L.mvx_vectors_size.argtypes = [P(AnalysisData)]
        L.mvx_vectors_size.restype = C.c_int
        _lib = L
        # developer / test switches: the library itself never reads the environment (mvx_debug_option is its one hook);
        # this TEST binding forwards the MVX_* variables the tools/ scripts use
        for env, opt in (("MVX_GENERAL", "general"), ("MVX_FAST_WPE", "fast_wpe"), ("MVX_NO_WPE2", "no_wpe2"),
                         ("MVX_NO_WPE3", "no_wpe3"), ("MVX_WPE3", "wpe3_u16"), ("MVX_FAST_CPW", "fast_cpw"), ("MVX_FAST_FLAGS", "fast_flags"), ("MVX_PAD_RUNS", "pad_runs"), ("MVX_SHADOW_PLANES", "shadow_planes"), ("MVX_DEGRAIN_XCD", "degrain_xcd"), ("MVX_DEGRAIN_SHADOW", "degrain_shadow"), ("MVX_CPW_SYNC", "cpw_sync"), ("MVX_LDS_MIN", "lds_min"), ("MVX_SUPER_ROWS_OFF", "super_rows_off"), ("MVX_ABLATE", "ablate"), ("MVX_FAST_K", "fast_k"), ("MVX_SPEC", "spec"), ("MVX_TEAM", "team"), ("MVX_FAST_LDS_MIN", "fast_lds_min"), ("MVX_SHADOW8", "shadow8")):
            if os.environ.get(env) is not None:
                L.mvx_debug_option(opt.encode(), int(os.environ[env]))
        if os.environ.get("MVX_CPW") == "1":
            L.mvx_debug_option(b"cpw1", 1)
    return _lib


def debug_option(name, value):
    """kernel-variant selection for tests / measurements (never changes results): see mvx_debug_option in mvtools_amd.h"""
    _check(lib().mvx_debug_option(name.encode(), int(value)))


def _u(v):
    return UNSET if v is None else int(v)


def _check(rc, err=None):
    if rc:
        msg = err.value.decode() if err is not None and err.value else lib().mvx_last_error().decode()
        raise MvtoolsError(msg)


def _torch():
    import torch
    if not torch.cuda.is_available():
        raise MvtoolsError("no HIP device visible: mvtools_amd has no CPU path")
    return torch


def arena_frames(n, plane_shapes, device="cuda", zero=True, slots=None):
    """n frames x len(plane_shapes) planes (rows, pitch_bytes) carved out of ONE uint8 device allocation (see Super.alloc).
    slots[p] > 1: plane p is followed by slots[p] - 1 further areas of its (256-byte rounded) size: room for the shadow planes."""
    torch = _torch()
    sizes = [r * p for r, p in plane_shapes]
    slots = slots or [1] * len(sizes)
    step = [(sz + 255) // 256 * 256 * k for sz, k in zip(sizes, slots)]
    total = n * sum(step)
    big = torch.zeros(total, dtype=torch.uint8, device=device) if zero else torch.empty(total, dtype=torch.uint8, device=device)
    out, o = [], 0
    for _ in range(n):
        row = []
        for (r, p), sz, st in zip(plane_shapes, sizes, step):
            row.append(big[o:o + sz].view(r, p))
            o += st
        out.append(row)
    return out


def _stream():
    return C.c_void_p(_torch().cuda.current_stream().cuda_stream)


def _pitches(planes):
    a = (C.c_ssize_t * 3)()
    for i, p in enumerate(planes):
        a[i] = p.stride(0)
    return a


# ---------------------------------------------------------------------------------------------- frame helpers

def plane_to_device(arr, pitch_align=256, device="cuda"):
    """numpy 2-D uint8/uint16 plane -> torch.uint8 [h, pitch] tensor (row stride = pitch bytes)."""
    torch = _torch()
    a = np.ascontiguousarray(arr)
    h, w = a.shape
    rowbytes = w * a.dtype.itemsize
    pitch = (rowbytes + pitch_align - 1) // pitch_align * pitch_align
    t = torch.zeros((h, pitch), dtype=torch.uint8, device=device)
    t[:, :rowbytes] = torch.from_numpy(a.view(np.uint8).reshape(h, rowbytes)).to(device)
    return t


def frame_to_device(planes, **kw):
    return [plane_to_device(p, **kw) for p in planes]


def plane_to_numpy(t, width, dtype):
    item = np.dtype(dtype).itemsize
    a = t[:, :width * item].contiguous().cpu().numpy()
    return a.view(dtype).reshape(t.shape[0], width)


class Super:
    """mv.Super -- MVSuper.c:140-275."""

    def __init__(self, width, height, bits=8, subsampling=(1, 1), gray=False, hpad=None, vpad=None, pel=None, levels=None,
                 chroma=None, sharp=None, rfilter=None, shadow=None):
        a = SuperArgs(width, height, bits, subsampling[0], subsampling[1], int(gray), _u(hpad), _u(vpad), _u(pel), _u(levels),
                      _u(chroma), _u(sharp), _u(rfilter))
        self.h = C.c_void_p()
        err = C.create_string_buffer(ERRLEN)
        _check(lib().mvx_super_create(C.byref(a), C.byref(self.h), err), err)
        self.info = SuperInfo()
        lib().mvx_super_get_info(self.h, C.byref(self.info))
        self.dtype = np.uint8 if bits <= 8 else np.uint16
        self.bps = 1 if bits <= 8 else 2
        self.nplanes = self.info.num_planes
        self.pitch = [((self.info.plane_width[p] * self.bps + 255) // 256) * 256 for p in range(self.nplanes)]
        # developer experiment (r4): row pitch as a number of 128-byte lines that is odd / a given residue -- how the rows of a block and the
        # sub-pel planes spread over the sets of the CU's L1.  MVX_PITCH_LINES="odd" | "<k>" (pitch = smallest number of lines >= the row that is == k mod 64)
        _pl = os.environ.get("MVX_PITCH_LINES")
        if _pl:
            for p in range(self.nplanes):
                n = (self.info.plane_width[p] * self.bps + 127) // 128
                if _pl == "odd":
                    n += 1 - (n & 1)
                else:
                    while n % 64 != int(_pl) % 64:
                        n += 1
                self.pitch[p] = n * 128
        # shadow planes (mvx_super_shadow_frames; clips of more than 8 bits): the luma plane is followed by its copy shifted by one
        # sample, the U plane by the UV-interleaved plane, so that the search only issues dword-aligned loads.  On by default
        # (MVX_SHADOW=0 / shadow=False: the plain layout).
        if shadow is None:
            shadow = os.environ.get("MVX_SHADOW", "1") != "0"
        self.shadow_stride = [(self.info.plane_height[p] * self.pitch[p] + 255) // 256 * 256 for p in range(self.nplanes)]
        extra = (C.c_size_t * 3)()
        if shadow:
            lib().mvx_super_shadow_bytes(self.h, (C.c_ssize_t * 3)(*(self.pitch + [0] * (3 - len(self.pitch)))), extra)
        self.shadow = any(extra)
        self.slots = [1 + (extra[p] + self.shadow_stride[p] - 1) // self.shadow_stride[p] for p in range(self.nplanes)]  # areas per plane: the plane + its shadow data

    def __del__(self):
        try:
            if self.h:
                lib().mvx_super_destroy(self.h)
        except Exception:
            pass

    def alloc(self, n=1, device="cuda"):
        """n zero-filled super frames (the library only ever writes the defined rectangles)."""
        torch = _torch()
        # ONE allocation for all frames, planes carved at 256-byte granularity: measured +10 % search throughput on 4K16
        # against one allocation per plane (the chains of a launch touch ~20 distinct plane regions each; a single arena
        # is mapped with large page fragments and keeps the TLBs effective).  MVX_ALLOC_ARENA=0 restores per-plane tensors.
        sizes = [self.info.plane_height[p] * self.pitch[p] for p in range(self.nplanes)]
        if os.environ.get("MVX_ALLOC_ARENA", "1") == "0":
            return [[torch.zeros(self.shadow_stride[p] * self.slots[p], dtype=torch.uint8, device=device)[:self.info.plane_height[p] * self.pitch[p]].view(self.info.plane_height[p], self.pitch[p])
                     for p in range(self.nplanes)] for _ in range(n)]
        return arena_frames(n, [(self.info.plane_height[p], self.pitch[p]) for p in range(self.nplanes)], device, slots=self.slots)

    def from_host(self, planes, device="cuda"):
        """a super frame given as host arrays (numpy planes of plane_height x >= plane_width samples, e.g. from another
        implementation) -> device super frame in this object's layout, shadow copies included"""
        torch = _torch()
        fr = self.alloc(1, device=device)[0]
        for p in range(self.nplanes):
            a = np.ascontiguousarray(planes[p][:, :self.info.plane_width[p]])
            fr[p][:, :a.shape[1] * a.dtype.itemsize] = torch.from_numpy(a.view(np.uint8).reshape(a.shape[0], -1)).to(device)
        self._shadows([fr])
        return fr

    def check_room(self, frames, what="super frame"):
        """With shadow planes the kernels write (Super) and read (Analyse) shadow_stride[p] * slots[p] bytes from every plane pointer:
        frames must come from alloc() / build() / from_host() of a Super with this layout.  A plain (height, pitch) tensor is refused
        here instead of being over-run on the device."""
        if not self.shadow:
            return
        seen = set()
        for fr in frames:
            if fr is None or id(fr) in seen:  # (a launch names every frame many times)
                continue
            seen.add(id(fr))
            for p in range(self.nplanes):
                t = fr[p]
                have = t.untyped_storage().nbytes() - t.storage_offset() * t.element_size()
                need = self.shadow_stride[p] * self.slots[p]
                if t.stride(0) != self.pitch[p] or have < need:
                    raise ValueError("%s plane %d: pitch %d with %d bytes behind its first sample, this Super's layout needs pitch %d and %d bytes "
                                     "(the plane and its shadow planes): allocate it with Super.alloc()" % (what, p, t.stride(0), have, self.pitch[p], need))

    def _shadows(self, out):
        """fills the shifted copies behind the planes of freshly built super frames"""
        if not self.shadow:
            return
        self.check_room(out, "output super frame")
        n = len(out)
        pl = (C.c_void_p * (3 * n))()
        for f in range(n):
            for p in range(self.nplanes):
                pl[f * 3 + p] = out[f][p].data_ptr()
        pad = lambda l: (C.c_ssize_t * 3)(*(list(l) + [0] * (3 - len(l))))
        _check(lib().mvx_super_shadow_frames(self.h, n, pl, pad(self.pitch), pad(self.shadow_stride), _stream()))

    def finest(self, super_frames, out=None):
        """mv.Finest(super) -- MVFinest.c: interleaved sub-pel planes of level 0, one output frame per super frame."""
        torch = _torch()
        n = len(super_frames)
        w, h = C.c_int32(), C.c_int32()
        lib().mvx_finest_size(self.h, C.byref(w), C.byref(h))
        i = self.info
        dims = [(h.value, w.value)] + [(h.value // i.yRatioUV, w.value // i.xRatioUV)] * 2
        pitch = [((dims[p][1] * self.bps + 255) // 256) * 256 for p in range(self.nplanes)]
        if out is None:
            out = arena_frames(n, [(dims[p][0], pitch[p]) for p in range(self.nplanes)], super_frames[0][0].device)
        src = (C.c_void_p * (3 * n))()
        dst = (C.c_void_p * (3 * n))()
        for f in range(n):
            for p in range(self.nplanes):
                src[f * 3 + p] = super_frames[f][p].data_ptr()
                dst[f * 3 + p] = out[f][p].data_ptr()
        pad = lambda l: (C.c_ssize_t * 3)(*(list(l) + [0] * (3 - len(l))))
        _check(lib().mvx_finest_frames(self.h, n, src, pad(self.pitch), dst, pad(pitch), _stream()))
        return out

    def pelclip_mode(self, pel_width, pel_height):
        """MVSuper.c:229-256: 0 = a pelclip is ignored (pel 1), 1 = plain, 2 = padded; raises on any other size."""
        mode = C.c_int32()
        err = C.create_string_buffer(ERRLEN)
        _check(lib().mvx_super_pelclip_mode(self.h, int(pel_width), int(pel_height), C.byref(mode), err), err)
        return mode.value

    def build(self, frames, out=None, pelclip=None, pelclip_size=None):
        """frames: list of device frames (list of plane tensors sharing pitches) -> list of super frames.
        pelclip: the matching frames of mv.Super's pelclip argument, pelclip_size = its (width, height)."""
        n = len(frames)
        if pelclip is not None:
            mode = self.pelclip_mode(*pelclip_size)
            if out is None:
                out = self.alloc(n, device=frames[0][0].device)
            src = (C.c_void_p * (3 * n))()
            pel = (C.c_void_p * (3 * n))()
            dst = (C.c_void_p * (3 * n))()
            for f in range(n):
                for p in range(self.nplanes):
                    src[f * 3 + p] = frames[f][p].data_ptr()
                    pel[f * 3 + p] = pelclip[f][p].data_ptr()
                    dst[f * 3 + p] = out[f][p].data_ptr()
            _check(lib().mvx_super_frames_pelclip(self.h, n, src, _pitches(frames[0]), pel, _pitches(pelclip[0]), mode, dst, _pitches(out[0]), _stream()))
            self._shadows(out)
            return out
        if out is None:
            out = self.alloc(n, device=frames[0][0].device)
        src = (C.c_void_p * (3 * n))()
        dst = (C.c_void_p * (3 * n))()
        for f in range(n):
            for p in range(self.nplanes):
                src[f * 3 + p] = frames[f][p].data_ptr()
                dst[f * 3 + p] = out[f][p].data_ptr()
                assert frames[f][p].stride(0) == frames[0][p].stride(0) and out[f][p].stride(0) == out[0][p].stride(0)
        if self.shadow:  # one call: the level-0 kernels write the shadow data of level 0 themselves
            self.check_room(out, "output super frame")
            pad = lambda l: (C.c_ssize_t * 3)(*(list(l) + [0] * (3 - len(l))))
            _check(lib().mvx_super_frames_shadow(self.h, n, src, _pitches(frames[0]), dst, _pitches(out[0]), pad(self.shadow_stride), _stream()))
        else:
            _check(lib().mvx_super_frames(self.h, n, src, _pitches(frames[0]), dst, _pitches(out[0]), _stream()))
        return out


class Analyse:
    """mv.Analyse -- MVAnalyse.c:267-635; keyword names are the reference's argument names."""

    def __init__(self, sup, num_frames=1 << 30, **kw):
        self.sup = sup
        a = AnalyseArgs(*([UNSET] * len(ANALYSE_ARGS)))
        for k, v in kw.items():
            k2 = {"lambda": "lambda_", "global": "global_"}.get(k, k)
            if k2 not in ANALYSE_ARGS:
                raise TypeError("Analyse: unknown argument " + k)
            if v is not None:
                setattr(a, k2, int(v))
        pitch = (C.c_ssize_t * 3)(*(sup.pitch + [0] * (3 - len(sup.pitch))))
        self.h = C.c_void_p()
        err = C.create_string_buffer(ERRLEN)
        _check(lib().mvx_analyse_create(C.byref(a), sup.h, int(num_frames), pitch, C.byref(self.h), err), err)
        self.ad = AnalysisData()
        lib().mvx_analyse_get_data(self.h, C.byref(self.ad))
        self.blob_size = lib().mvx_analyse_blob_size(self.h)
        if sup.shadow:  # super frames from sup.alloc / sup.build carry their shifted copies
            _check(lib().mvx_analyse_set_ref_shadow(self.h, (C.c_ssize_t * 3)(*(sup.shadow_stride + [0] * (3 - len(sup.shadow_stride))))))

    def __del__(self):
        try:
            if self.h:
                lib().mvx_analyse_destroy(self.h)
        except Exception:
            pass

    def alloc_blobs(self, n, device="cuda"):
        torch = _torch()
        stride = (self.blob_size + 255) // 256 * 256
        buf = torch.zeros((n, stride), dtype=torch.uint8, device=device)
        return [buf[i, :self.blob_size] for i in range(n)]

    def run(self, jobs, blobs=None, field_shift=0):
        """jobs: list of (src_super_frame, ref_super_frame_or_None) -> list of device blobs (MVTools_vectors)."""
        n = len(jobs)
        if blobs is None:
            blobs = self.alloc_blobs(n, device=jobs[0][0][0].device)
        arr = (AnalyseJob * n)()
        self.sup.check_room([f for job in jobs for f in job], "Analyse input")  # (the search reads the shadow planes behind every plane)
        for i, (s, r) in enumerate(jobs):
            for p in range(self.sup.nplanes):
                assert s[p].stride(0) == self.sup.pitch[p]
                arr[i].src[p] = s[p].data_ptr()
                arr[i].ref[p] = r[p].data_ptr() if r is not None else None
            arr[i].blob = blobs[i].data_ptr()
            arr[i].field_shift = field_shift
        _check(lib().mvx_analyse_frames(self.h, n, arr, _stream()))
        return blobs


class Degrain:
    """mv.Degrain1..6 -- MVDegrains.cpp:511-809.  analysis_data = the vector clips' MVTools_MVAnalysisData."""

    def __init__(self, radius, sup, analysis_data, src_pitch, dst_pitch=None, thsad=None, thsadc=None, plane=None, limit=None,
                 limitc=None, thscd1=None, thscd2=None):
        self.sup = sup
        self.radius = radius
        a = DegrainArgs(radius, _u(thsad), _u(thsadc), _u(plane), _u(limit), _u(limitc), _u(thscd1), _u(thscd2))
        ad = AnalysisData.from_buffer_copy(bytes(analysis_data))
        dst_pitch = dst_pitch or src_pitch
        pad = lambda l: (C.c_ssize_t * 3)(*(list(l) + [0] * (3 - len(l))))
        self.h = C.c_void_p()
        err = C.create_string_buffer(ERRLEN)
        _check(lib().mvx_degrain_create(C.byref(a), C.byref(ad), sup.h, pad(src_pitch), pad(sup.pitch), pad(dst_pitch), C.byref(self.h), err), err)
        self.src_pitch, self.dst_pitch = list(src_pitch), list(dst_pitch)
        self.ref_shadow = bool(sup.shadow and sup.slots[0] > 1)
        if self.ref_shadow:  # super frames from sup.alloc / sup.build carry the shifted copy of their luma plane (clips of more than 8 bits)
            _check(lib().mvx_degrain_set_ref_shadow(self.h, (C.c_ssize_t * 3)(*(sup.shadow_stride + [0] * (3 - len(sup.shadow_stride))))))

    def __del__(self):
        try:
            if self.h:
                lib().mvx_degrain_destroy(self.h)
        except Exception:
            pass

    def run(self, jobs, out=None):
        """jobs: list of (src_frame, [ref_super or None]*2r, [blob]*2r) ordered mvbw, mvfw, mvbw2, mvfw2, ..."""
        torch = _torch()
        n = len(jobs)
        if out is None:
            out = [[torch.empty_like(p) for p in j[0]] for j in jobs]
        arr = (DegrainJob * n)()
        if self.ref_shadow:  # (blocks at odd sample positions are read from the shifted luma copy BEHIND every reference plane: a plain tensor would be over-run)
            self.sup.check_room([r for _, refs, _ in jobs for r in refs], "Degrain reference")
        for i, (src, refs, blobs) in enumerate(jobs):
            for p in range(self.sup.nplanes):
                assert src[p].stride(0) == self.src_pitch[p] and out[i][p].stride(0) == self.dst_pitch[p]
                arr[i].src[p] = src[p].data_ptr()
                arr[i].dst[p] = out[i][p].data_ptr()
            for r in range(2 * self.radius):
                if refs[r] is not None:
                    for p in range(self.sup.nplanes):
                        arr[i].refs[r][p] = refs[r][p].data_ptr()
                arr[i].blobs[r] = blobs[r].data_ptr()
        _check(lib().mvx_degrain_frames(self.h, n, arr, _stream()))
        return out


class Compensate:
    """mv.Compensate -- MVCompensate.c:419-575."""

    def __init__(self, sup, analysis_data, dst_pitch=None, scbehavior=None, thsad=None, time=100.0, thscd1=None, thscd2=None, fields=None):
        self.sup = sup
        a = CompensateArgs(_u(scbehavior), _u(thsad), float(time), _u(thscd1), _u(thscd2), _u(fields))
        ad = AnalysisData.from_buffer_copy(bytes(analysis_data))
        i = sup.info
        if dst_pitch is None:
            w = [i.width] + [i.width // i.xRatioUV] * 2
            dst_pitch = [((w[p] * sup.bps + 255) // 256) * 256 for p in range(sup.nplanes)]
        pad = lambda l: (C.c_ssize_t * 3)(*(list(l) + [0] * (3 - len(l))))
        self.h = C.c_void_p()
        err = C.create_string_buffer(ERRLEN)
        _check(lib().mvx_compensate_create(C.byref(a), C.byref(ad), sup.h, pad(sup.pitch), pad(dst_pitch), C.byref(self.h), err), err)
        self.dst_pitch = list(dst_pitch)

    def __del__(self):
        try:
            if self.h:
                lib().mvx_compensate_destroy(self.h)
        except Exception:
            pass

    def run(self, jobs, out=None):
        """jobs: list of (src_super, ref_super_or_None, blob[, field_shift])."""
        torch = _torch()
        i = self.sup.info
        n = len(jobs)
        hs = [i.height] + [i.height // i.yRatioUV] * 2
        if out is None:
            dev = jobs[0][0][0].device
            out = [[torch.zeros((hs[p], self.dst_pitch[p]), dtype=torch.uint8, device=dev) for p in range(self.sup.nplanes)] for _ in range(n)]
        arr = (CompensateJob * n)()
        for k, job in enumerate(jobs):
            s, r, blob = job[:3]
            for p in range(self.sup.nplanes):
                arr[k].src_super[p] = s[p].data_ptr()
                arr[k].ref_super[p] = r[p].data_ptr() if r is not None else None
                arr[k].dst[p] = out[k][p].data_ptr()
            arr[k].blob = blob.data_ptr()
            arr[k].field_shift = int(job[3]) if len(job) > 3 else 0
        _check(lib().mvx_compensate_frames(self.h, n, arr, _stream()))
        return out


class BlockFPS:
    """mv.BlockFPS(clip, super, mvbw, mvfw, num, den, mode, ml, blend, thscd1, thscd2) -- MVBlockFPS.c:741-1014.
    `fps_num / fps_den` is the input clip's frame rate; `clip_pitch` the row pitch of its device planes."""

    def __init__(self, sup, ad_bw, ad_fw, num_frames, clip_pitch, fps_num=24, fps_den=1, num=None, den=None, mode=None, ml=100.0, blend=None, thscd1=None,
                 thscd2=None):
        self.sup = sup
        a = BlockFPSArgs(_u(num), _u(den), _u(mode), float(ml), _u(blend), _u(thscd1), _u(thscd2))
        bw = AnalysisData.from_buffer_copy(bytes(ad_bw))
        fw = AnalysisData.from_buffer_copy(bytes(ad_fw))
        pad = lambda l: (C.c_ssize_t * 3)(*(list(l) + [0] * (3 - len(l))))
        self.h = C.c_void_p()
        self.pitch = list(clip_pitch)
        err = C.create_string_buffer(ERRLEN)
        _check(lib().mvx_blockfps_create(C.byref(a), C.byref(bw), C.byref(fw), sup.h, int(num_frames), int(fps_num), int(fps_den), pad(sup.pitch), pad(clip_pitch),
                                         pad(clip_pitch), C.byref(self.h), err), err)
        self.in_frames = int(num_frames)
        info = BlockFPSInfo()
        lib().mvx_blockfps_get_info(self.h, C.byref(info))
        self.num_frames, self.fps_num, self.fps_den = info.num_frames, info.fps_num, info.fps_den

    def __del__(self):
        try:
            if self.h:
                lib().mvx_blockfps_destroy(self.h)
        except Exception:
            pass

    def map(self, n):
        a, b, c = C.c_int(), C.c_int(), C.c_int()
        lib().mvx_blockfps_map(self.h, n, C.byref(a), C.byref(b), C.byref(c))
        return a.value, b.value, c.value

    def run(self, frames_out, clip, supers, blobs_bw, blobs_fw, out=None):
        """frames_out: output frame numbers; clip / supers: device frames of the input clip and its super clip; blobs_*: per
        input frame device blobs of the two vector clips (mvbw at n, mvfw at n)."""
        torch = _torch()
        n = len(frames_out)
        if out is None:
            out = arena_frames(n, [tuple(p.shape) for p in clip[0]], clip[0][0].device, zero=False)
        arr = (BlockFPSJob * n)()
        last = self.in_frames - 1
        for k, fo in enumerate(frames_out):
            nl, nr, t = self.map(fo)
            arr[k].time256 = t
            good = nl < self.in_frames and nr < self.in_frames
            L, R = clip[min(nl, last)], clip[min(nr, last)]
            for p in range(self.sup.nplanes):
                arr[k].clip_left[p] = L[p].data_ptr()
                arr[k].clip_right[p] = R[p].data_ptr()
                arr[k].dst[p] = out[k][p].data_ptr()
                if good:
                    arr[k].src_super[p] = supers[nl][p].data_ptr()
                    arr[k].ref_super[p] = supers[nr][p].data_ptr()
            if good:
                arr[k].blob_fw = blobs_fw[nr].data_ptr()
                arr[k].blob_bw = blobs_bw[nl].data_ptr()
        _check(lib().mvx_blockfps_frames(self.h, n, arr, _stream()))
        return out


class Recalculate:
    """mv.Recalculate(super, vectors, thsad, smooth, blksize, ...) -- MVRecalculate.c:263-545."""

    def __init__(self, sup, vectors_ad, **kw):
        self.sup = sup
        a = RecalculateArgs(*([UNSET] * len(RECALC_ARGS)))
        for k, v in kw.items():
            k2 = {"lambda": "lambda_"}.get(k, k)
            if k2 not in RECALC_ARGS:
                raise TypeError("Recalculate: unknown argument " + k)
            if v is not None:
                setattr(a, k2, int(v))
        old = AnalysisData.from_buffer_copy(bytes(vectors_ad))
        pad = lambda l: (C.c_ssize_t * 3)(*(list(l) + [0] * (3 - len(l))))
        self.h = C.c_void_p()
        err = C.create_string_buffer(ERRLEN)
        _check(lib().mvx_recalculate_create(C.byref(a), sup.h, C.byref(old), pad(sup.pitch), C.byref(self.h), err), err)
        self.ad = AnalysisData()
        lib().mvx_recalculate_get_data(self.h, C.byref(self.ad))
        self.blob_size = lib().mvx_recalculate_blob_size(self.h)

    def __del__(self):
        try:
            if self.h:
                lib().mvx_recalculate_destroy(self.h)
        except Exception:
            pass

    def run(self, jobs, blobs=None):
        """jobs: list of (src_super, ref_super_or_None, old_blob) -> list of device blobs."""
        torch = _torch()
        n = len(jobs)
        if blobs is None:
            stride = (self.blob_size + 255) // 256 * 256
            buf = torch.zeros((n, stride), dtype=torch.uint8, device=jobs[0][0][0].device)
            blobs = [buf[i, :self.blob_size] for i in range(n)]
        arr = (RecalculateJob * n)()
        for i, (s, r, ob) in enumerate(jobs):
            for p in range(self.sup.nplanes):
                arr[i].src[p] = s[p].data_ptr()
                arr[i].ref[p] = r[p].data_ptr() if r is not None else None
            arr[i].old_blob = ob.data_ptr()
            arr[i].blob = blobs[i].data_ptr()
        _check(lib().mvx_recalculate_frames(self.h, n, arr, _stream()))
        return blobs


def scdetect(analysis_data, blobs, thscd1=None, thscd2=None):
    """mv.SCDetection's decision per frame (MVSCDetection.c:43-73): list of 0/1 = value of _SceneChangePrev/_SceneChangeNext."""
    n = len(blobs)
    ad = AnalysisData.from_buffer_copy(bytes(analysis_data))
    ptrs = (C.c_void_p * n)(*[b.data_ptr() for b in blobs])
    out = (C.c_int32 * n)()
    err = C.create_string_buffer(ERRLEN)
    _check(lib().mvx_scdetect(C.byref(ad), _u(thscd1), _u(thscd2), n, ptrs, out, _stream(), err), err)
    return list(out)
