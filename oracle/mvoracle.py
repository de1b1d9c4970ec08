"""ctypes binding of the CPU oracle (oracle/libmvoracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg,
never by the product package.  The interface mirrors the reference filters: Super / Analyse / Degrain /
Compensate objects built from the same arguments as mv.Super / mv.Analyse / mv.DegrainN / mv.Compensate
(/root/reference/src/MVSuper.c:279-291, MVAnalyse.c:639-671, MVDegrains.cpp:813-932, MVCompensate.c:579-592).
Frames are lists of numpy planes (uint8 or uint16, C-contiguous 2-D arrays whose row stride is the pitch).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
UNSET = -2147483648
ERRLEN = 256


def build(force=False):
    so = os.path.join(_HERE, "libmvoracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("mvo_super.c", "mvo_analyse.c", "mvo_degrain.c", "mvoracle.h", "mvo_internal.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "libmvoracle.so"], stdout=subprocess.DEVNULL)
    return so


class AnalysisData(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        "nMagicKey", "nVersion", "nBlkSizeX", "nBlkSizeY", "nPel", "nLvCount", "nDeltaFrame", "isBackward", "nCPUFlags",
        "nMotionFlags", "nWidth", "nHeight", "nOverlapX", "nOverlapY", "nBlkX", "nBlkY", "bitsPerSample", "yRatioUV",
        "xRatioUV", "nHPadding", "nVPadding")]


class SuperS(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        "width", "height", "bits", "xRatioUV", "yRatioUV", "gray", "hpad", "vpad", "pel", "levels", "chroma", "sharp",
        "rfilter", "modeYUV", "superWidth", "superHeight")]


ANALYSE_ARGS = ("blksize", "blksizev", "levels", "search", "searchparam", "pelsearch", "isb", "lambda_", "chroma", "delta",
                "truemotion", "lsad", "plevel", "global_", "pnew", "pzero", "pglobal", "overlap", "overlapv", "divide", "badsad",
                "badrange", "opt", "meander", "trymany", "fields", "tff", "search_coarse", "dct")


class AnalyseArgs(C.Structure):
    _fields_ = [(n, C.c_int) for n in ANALYSE_ARGS]


class AnalyseS(C.Structure):
    _fields_ = [("ad", AnalysisData)] + [(n, C.c_int) for n in (
        "searchType", "searchTypeCoarse", "nSearchParam", "nPelSearch", "nLambda", "lsad", "pnew", "plevel", "global_",
        "pglobal", "pzero", "divideExtra", "badrange", "meander", "tryMany", "dctmode", "chroma", "fields", "tff",
        "tff_exists", "opt")] + [("badSAD", C.c_int64)] + [(n, C.c_int) for n in (
            "nSuperLevels", "nSuperHPad", "nSuperVPad", "nSuperPel", "nSuperModeYUV", "numFrames")]


class DegrainS(C.Structure):
    _fields_ = [("radius", C.c_int), ("ad", AnalysisData), ("thSAD", C.c_int64 * 3), ("nSCD1", C.c_int64), ("nSCD2", C.c_int),
                ("nLimit", C.c_int * 3), ("process", C.c_int * 3)] + [(n, C.c_int) for n in (
                    "nSuperHPad", "nSuperVPad", "nSuperPel", "nSuperModeYUV", "nSuperLevels", "bits", "numPlanes", "xSubUV",
                    "ySubUV")] + [(n, C.c_int * 3) for n in (
                        "nWidth", "nHeight", "nOverlapX", "nOverlapY", "nBlkSizeX", "nBlkSizeY", "nWidth_B", "nHeight_B")]


class CompensateS(C.Structure):
    _fields_ = [("ad", AnalysisData), ("thSAD", C.c_int64), ("nSCD1", C.c_int64), ("nSCD2", C.c_int)] + [
        (n, C.c_int) for n in ("scBehavior", "time256", "fields", "nSuperHPad", "nSuperVPad", "nSuperPel", "nSuperModeYUV",
                               "nSuperLevels", "bits", "numPlanes")]


class BlockFPSS(C.Structure):
    _fields_ = [("bw", AnalysisData), ("fw", AnalysisData), ("mode", C.c_int), ("blend", C.c_int), ("ml", C.c_double), ("thscd1", C.c_int64),
                ("thscd2", C.c_int), ("fa", C.c_int64), ("fb", C.c_int64), ("outFpsNum", C.c_int64), ("outFpsDen", C.c_int64)] + [
        (n, C.c_int) for n in ("inFrames", "outFrames", "nSuperHPad", "nSuperVPad", "nSuperPel", "nSuperModeYUV", "nSuperLevels", "bits",
                               "nBlkXP", "nBlkYP", "nWidthP", "nHeightP", "nWidthPUV", "nHeightPUV", "nPitchY", "nPitchUV")]


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        P = C.POINTER
        u8p = C.c_void_p
        _lib.mvo_plane_height_luma.restype = C.c_int
        _lib.mvo_plane_width_luma.restype = C.c_int
        _lib.mvo_plane_super_offset.restype = C.c_uint
        _lib.mvo_analyse_blob_size.restype = C.c_int
        _lib.mvo_blob_is_usable.restype = C.c_int
        _lib.mvo_blob_is_usable.argtypes = [P(AnalysisData), u8p, C.c_int64, C.c_int]
        _lib.mvo_sad.restype = C.c_uint
        _lib.mvo_sad.argtypes = [C.c_int, C.c_int, C.c_int, u8p, C.c_ssize_t, u8p, C.c_ssize_t]
        _lib.mvo_satd.restype = C.c_uint
        _lib.mvo_satd.argtypes = [C.c_int, C.c_int, C.c_int, u8p, C.c_ssize_t, u8p, C.c_ssize_t]
        _lib.mvo_overlaps.argtypes = [C.c_int, C.c_int, C.c_int, u8p, C.c_ssize_t, u8p, C.c_ssize_t, u8p, C.c_ssize_t]
        _lib.mvo_to_pixels.argtypes = [C.c_int, u8p, C.c_int, u8p, C.c_int, C.c_int, C.c_int]
        _lib.mvo_refine_plane.argtypes = [C.c_int, C.c_int, u8p, u8p, C.c_ssize_t, C.c_ssize_t, C.c_ssize_t]
        _lib.mvo_average2.argtypes = [C.c_int, u8p, u8p, u8p, C.c_ssize_t, C.c_ssize_t, C.c_ssize_t]
        _lib.mvo_over_init.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_int]
        _lib.mvo_fnv1a.restype = C.c_uint32
        _lib.mvo_fnv1a.argtypes = [u8p, C.c_size_t]
        _lib.mvo_degrain_init.argtypes = [P(DegrainS), C.c_int, P(AnalysisData), P(SuperS), C.c_int64, C.c_int64, C.c_int,
                                          C.c_int, C.c_int, C.c_int64, C.c_int, C.c_char_p]
        _lib.mvo_compensate_init.argtypes = [P(CompensateS), P(AnalysisData), P(SuperS), C.c_int, C.c_int64, C.c_double,
                                             C.c_int64, C.c_int, C.c_char_p]
        _lib.mvo_blockfps_init.argtypes = [P(BlockFPSS), P(AnalysisData), P(AnalysisData), P(SuperS), C.c_int, C.c_int64, C.c_int64, C.c_int64,
                                           C.c_int64, C.c_int, C.c_double, C.c_int, C.c_int64, C.c_int, C.c_char_p]
        _lib.mvo_blockfps_map.argtypes = [P(BlockFPSS), C.c_int, P(C.c_int), P(C.c_int), P(C.c_int)]
        _lib.mvo_blockfps_frame.restype = C.c_int
        _lib.mvo_blockfps_frame.argtypes = [P(BlockFPSS), C.c_int] + [C.c_void_p] * 12
        _lib.mvo_resize_tables.argtypes = [u8p, u8p, C.c_int, C.c_int]
    return _lib


class OracleError(Exception):
    pass


def field_shift(fields, pel, n, nref, src_field=-1, ref_field=-1, tff=-1):
    """MVAnalyse.c:135-176 / MVCompensate.c:188-225.  Returns (shift, missing)."""
    miss = C.c_int(0)
    v = lib().mvo_field_shift(int(fields), int(pel), int(n), int(nref), int(src_field), int(ref_field), int(tff), C.byref(miss))
    return v, bool(miss.value)


def _u(v):
    return UNSET if v is None else int(v)


def _planes(frame):
    """-> (void*[3], int[3]) for a list of 1 or 3 numpy planes."""
    ptrs = (C.c_void_p * 3)()
    pitch = (C.c_int * 3)()
    for i, p in enumerate(frame):
        assert p.flags["C_CONTIGUOUS"] or p.strides[1] == p.itemsize
        ptrs[i] = p.ctypes.data
        pitch[i] = p.strides[0]
    return ptrs, pitch


def fnv1a(arr):
    a = np.ascontiguousarray(arr)
    return lib().mvo_fnv1a(a.ctypes.data, a.nbytes)


class Super:
    """mv.Super(clip, hpad, vpad, pel, levels, chroma, sharp, rfilter) -- MVSuper.c:140-275."""

    def __init__(self, width, height, bits=8, subsampling=(1, 1), gray=False, hpad=None, vpad=None, pel=None, levels=None,
                 chroma=None, sharp=None, rfilter=None):
        self.s = SuperS()
        err = C.create_string_buffer(ERRLEN)
        rc = lib().mvo_super_init(C.byref(self.s), width, height, bits, subsampling[0], subsampling[1], int(gray), _u(hpad),
                                  _u(vpad), _u(pel), _u(levels), _u(chroma), _u(sharp), _u(rfilter), err)
        if rc:
            raise OracleError(err.value.decode())
        self.dtype = np.uint8 if bits <= 8 else np.uint16
        self.nplanes = 1 if gray else 3

    def plane_shape(self, p):
        s = self.s
        if p == 0:
            return s.superHeight, s.superWidth
        return s.superHeight // s.yRatioUV, s.superWidth // s.xRatioUV

    def alloc(self, pitch_align=64):
        out = []
        for p in range(self.nplanes):
            h, w = self.plane_shape(p)
            item = np.dtype(self.dtype).itemsize
            pitch = (w * item + pitch_align - 1) // pitch_align * pitch_align
            buf = np.zeros((h, pitch // item), dtype=self.dtype)
            out.append(buf[:, :w] if False else buf)  # keep full pitch so row stride == pitch
        return out

    def finest(self, super_frame):
        """mv.Finest(super) -- MVFinest.c: the interleaved sub-pel planes of level 0."""
        w, h = C.c_int(), C.c_int()
        lib().mvo_finest_size(C.byref(self.s), C.byref(w), C.byref(h))
        dst = [np.zeros((h.value, w.value), dtype=self.dtype)]
        if self.nplanes == 3:
            dst += [np.zeros((h.value // self.s.yRatioUV, w.value // self.s.xRatioUV), dtype=self.dtype) for _ in range(2)]
        sp, spitch = _planes(super_frame)
        dp, dpitch = _planes(dst)
        lib().mvo_finest_frame(C.byref(self.s), sp, spitch, dp, dpitch)
        return dst

    def frame(self, src):
        dst = self.alloc()
        sp, spitch = _planes(src)
        dp, dpitch = _planes(dst)
        lib().mvo_super_frame(C.byref(self.s), sp, spitch, dp, dpitch)
        return dst

    def pelclip_mode(self, pel_width, pel_height):
        """MVSuper.c:229-256: 0 = pelclip ignored, 1 = plain, 2 = padded; raises on other sizes."""
        err = C.create_string_buffer(ERRLEN)
        m = lib().mvo_super_pelclip_mode(C.byref(self.s), int(pel_width), int(pel_height), err)
        if m < 0:
            raise OracleError(err.value.decode())
        return m

    def frame_pelclip(self, src, pelclip):
        """mv.Super(clip, pelclip=...) for one frame (MVSuper.c:91-102, MVFrame.cpp:1529-1631)."""
        mode = self.pelclip_mode(pelclip[0].shape[1], pelclip[0].shape[0])
        dst = self.alloc()
        sp, spitch = _planes(src)
        pp, ppitch = _planes(pelclip)
        dp, dpitch = _planes(dst)
        lib().mvo_super_frame_pelclip(C.byref(self.s), sp, spitch, pp, ppitch, mode, dp, dpitch)
        return dst

    def defined_regions(self):
        """[(plane, level, pelplane, y0, x0, h, w)] of every defined rectangle of the super frame."""
        s = self.s
        L = lib()
        out = []
        for p in range(self.nplanes):
            xr, yr = (s.xRatioUV, s.yRatioUV) if p else (1, 1)
            hp, vp = s.hpad // xr, s.vpad // yr
            h0 = s.height // yr
            for lv in range(s.levels):
                wl = L.mvo_plane_width_luma(s.width, lv, s.xRatioUV, s.hpad) // xr
                hl = L.mvo_plane_height_luma(s.height, lv, s.yRatioUV, s.vpad) // yr
                npl = s.pel * s.pel if lv == 0 else 1
                # row offset of the level inside the plane, in rows (pitch-independent)
                off_rows = L.mvo_plane_super_offset(p, h0, lv, s.pel, vp, 1, s.yRatioUV)
                for k in range(npl):
                    out.append((p, lv, k, off_rows + k * (hl + 2 * vp), 0, hl + 2 * vp, wl + 2 * hp))
        return out


class Analyse:
    """mv.Analyse(super, ...) -- MVAnalyse.c:267-635.  Keyword names are the reference's argument names."""

    def __init__(self, sup, num_frames=1 << 30, **kw):
        self.sup = sup
        a = AnalyseArgs()
        lib().mvo_analyse_args_default(C.byref(a))
        for k, v in kw.items():
            k2 = {"lambda": "lambda_", "global": "global_"}.get(k, k)
            if k2 not in ANALYSE_ARGS:
                raise TypeError("Analyse: unknown argument " + k)
            if v is not None:
                setattr(a, k2, int(v))
        self.d = AnalyseS()
        err = C.create_string_buffer(ERRLEN)
        if lib().mvo_analyse_init(C.byref(self.d), C.byref(a), C.byref(sup.s), int(num_frames), err):
            raise OracleError(err.value.decode())
        self.blob_size = lib().mvo_analyse_blob_size(C.byref(self.d))

    @property
    def ad(self):
        """the analysis data readers see (the divided geometry when divide > 0)"""
        if self.d.divideExtra:
            out = AnalysisData()
            lib().mvo_analysis_data_divided(C.byref(self.d.ad), C.byref(out))
            return out
        return self.d.ad

    def frame(self, src_super, ref_super, field_shift=0):
        """ref_super=None -> invalid (default) blob, as for frames too close to the clip boundary."""
        blob = np.zeros(self.blob_size, dtype=np.uint8)
        sp, spitch = _planes(src_super)
        if ref_super is None:
            lib().mvo_analyse_frame(C.byref(self.d), sp, spitch, None, None, 0, C.c_void_p(blob.ctypes.data))
        else:
            rp, rpitch = _planes(ref_super)
            lib().mvo_analyse_frame(C.byref(self.d), sp, spitch, rp, rpitch, int(field_shift), C.c_void_p(blob.ctypes.data))
        return blob


def _alloc_like(src):
    return [np.zeros_like(p) for p in src]


class Degrain:
    """mv.DegrainN(clip, super, mvbw, mvfw, ..., thsad, thsadc, plane, limit, limitc, thscd1, thscd2) -- MVDegrains.cpp:511-809."""

    def __init__(self, radius, sup, analysis_data, thsad=None, thsadc=None, plane=None, limit=None, limitc=None, thscd1=None, thscd2=None):
        self.d = DegrainS()
        err = C.create_string_buffer(ERRLEN)
        ad = AnalysisData.from_buffer_copy(bytes(analysis_data))
        if lib().mvo_degrain_init(C.byref(self.d), radius, C.byref(ad), C.byref(sup.s), _u(thsad), _u(thsadc), _u(plane),
                                  _u(limit), _u(limitc), _u(thscd1), _u(thscd2), err):
            raise OracleError(err.value.decode())
        self.radius = radius

    def frame(self, src, ref_supers, blobs):
        """ref_supers[r] / blobs[r] ordered mvbw, mvfw, mvbw2, mvfw2, ...; ref_supers[r] may be None."""
        n = 2 * self.radius
        dst = _alloc_like(src)
        sp, spitch = _planes(src)
        dp, dpitch = _planes(dst)
        refs = ((C.c_void_p * 3) * n)()
        rpitch = ((C.c_int * 3) * n)()
        keep = []
        for r in range(n):
            if ref_supers[r] is not None:
                for i, p in enumerate(ref_supers[r]):
                    refs[r][i] = p.ctypes.data
                    rpitch[r][i] = p.strides[0]
        bl = (C.c_void_p * n)()
        for r in range(n):
            b = np.ascontiguousarray(blobs[r])
            keep.append(b)
            bl[r] = b.ctypes.data
        lib().mvo_degrain_frame(C.byref(self.d), sp, spitch, refs, rpitch, bl, dp, dpitch)
        return dst


class Compensate:
    """mv.Compensate(clip, super, vectors, scbehavior, thsad, time, thscd1, thscd2) -- MVCompensate.c:419-575."""

    def __init__(self, sup, analysis_data, scbehavior=None, thsad=None, time=100.0, thscd1=None, thscd2=None):
        self.d = CompensateS()
        self.sup = sup
        err = C.create_string_buffer(ERRLEN)
        ad = AnalysisData.from_buffer_copy(bytes(analysis_data))
        if lib().mvo_compensate_init(C.byref(self.d), C.byref(ad), C.byref(sup.s), _u(scbehavior), _u(thsad), float(time),
                                     _u(thscd1), _u(thscd2), err):
            raise OracleError(err.value.decode())

    def frame(self, src_super, ref_super, blob, field_shift=0):
        s = self.sup.s
        dst = [np.zeros((s.height, s.width), dtype=self.sup.dtype)]
        if self.sup.nplanes == 3:
            dst += [np.zeros((s.height // s.yRatioUV, s.width // s.xRatioUV), dtype=self.sup.dtype) for _ in range(2)]
        sp, spitch = _planes(src_super)
        dp, dpitch = _planes(dst)
        b = np.ascontiguousarray(blob)
        if ref_super is None:
            lib().mvo_compensate_frame(C.byref(self.d), sp, spitch, None, None, C.c_void_p(b.ctypes.data), dp, dpitch, int(field_shift))
        else:
            rp, rpitch = _planes(ref_super)
            lib().mvo_compensate_frame(C.byref(self.d), sp, spitch, rp, rpitch, C.c_void_p(b.ctypes.data), dp, dpitch, int(field_shift))
        return dst


class BlockFPS:
    """mv.BlockFPS(clip, super, mvbw, mvfw, num, den, mode, ml, blend, thscd1, thscd2) -- MVBlockFPS.c:741-1014.
    The clip's frame rate is fps_num / fps_den.  Parity of this filter is unpinned (see mvo_blockfps.c)."""

    def __init__(self, sup, ad_bw, ad_fw, num_frames, fps_num=24, fps_den=1, num=None, den=None, mode=None, ml=100.0, blend=None, thscd1=None, thscd2=None):
        self.d = BlockFPSS()
        self.sup = sup
        err = C.create_string_buffer(ERRLEN)
        bw = AnalysisData.from_buffer_copy(bytes(ad_bw))
        fw = AnalysisData.from_buffer_copy(bytes(ad_fw))
        if lib().mvo_blockfps_init(C.byref(self.d), C.byref(bw), C.byref(fw), C.byref(sup.s), int(num_frames), int(fps_num), int(fps_den), _u(num), _u(den),
                                   _u(mode), float(ml), _u(blend), _u(thscd1), _u(thscd2), err):
            raise OracleError(err.value.decode())
        self.num_frames = self.d.outFrames

    def map(self, n):
        a, b, c = C.c_int(), C.c_int(), C.c_int()
        lib().mvo_blockfps_map(C.byref(self.d), n, C.byref(a), C.byref(b), C.byref(c))
        return a.value, b.value, c.value

    def frame(self, n, clip, supers, blobs_bw, blobs_fw):
        """output frame n from the input clip (list of frames), its super frames and the two vector clips' blobs (per input frame)."""
        nleft, nright, t = self.map(n)
        last = self.d.inFrames - 1
        if t == 0:
            return [p.copy() for p in clip[min(nleft, last)]]
        if t == 256:
            return [p.copy() for p in clip[min(nright, last)]]
        L, R = clip[min(nleft, last)], clip[min(nright, last)]
        dst = _alloc_like(L)
        dp, dpitch = _planes(dst)
        lp, lpitch = _planes(L)
        rp, rpitch = _planes(R)
        good = nleft < self.d.inFrames and nright < self.d.inFrames
        if good:
            sp, spitch = _planes(supers[nleft])
            fp, fpitch = _planes(supers[nright])
            bF = np.ascontiguousarray(blobs_fw[nright])
            bB = np.ascontiguousarray(blobs_bw[nleft])
            rc = lib().mvo_blockfps_frame(C.byref(self.d), t, sp, spitch, fp, fpitch, C.c_void_p(bF.ctypes.data), C.c_void_p(bB.ctypes.data), lp, lpitch, rp, rpitch, dp, dpitch)
        else:
            rc = lib().mvo_blockfps_frame(C.byref(self.d), t, None, None, None, None, None, None, lp, lpitch, rp, rpitch, dp, dpitch)
        return [p.copy() for p in L] if rc == 1 else dst


class RecalculateArgs(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("thsad", "smooth", "blksize", "blksizev", "search", "searchparam", "lambda_", "chroma", "truemotion", "pnew", "overlap",
                                         "overlapv", "divide", "meander", "dct")]


class RecalculateS(C.Structure):
    _fields_ = [("an", AnalyseS), ("old", AnalysisData), ("thSAD", C.c_int64), ("smooth", C.c_int)]


class Recalculate:
    """mv.Recalculate(super, vectors, thsad, smooth, blksize, ...) -- MVRecalculate.c:263-545."""

    def __init__(self, sup, vectors_ad, **kw):
        self.sup = sup
        a = RecalculateArgs()
        lib().mvo_recalculate_args_default(C.byref(a))
        for k, v in kw.items():
            k2 = {"lambda": "lambda_"}.get(k, k)
            if k2 not in [n for n, _ in RecalculateArgs._fields_]:
                raise TypeError("Recalculate: unknown argument " + k)
            if v is not None:
                setattr(a, k2, int(v))
        self.d = RecalculateS()
        old = AnalysisData.from_buffer_copy(bytes(vectors_ad))
        err = C.create_string_buffer(ERRLEN)
        if lib().mvo_recalculate_init(C.byref(self.d), C.byref(a), C.byref(sup.s), C.byref(old), err):
            raise OracleError(err.value.decode())
        self.blob_size = lib().mvo_recalculate_blob_size(C.byref(self.d))

    @property
    def ad(self):
        if self.d.an.divideExtra:
            out = AnalysisData()
            lib().mvo_analysis_data_divided(C.byref(self.d.an.ad), C.byref(out))
            return out
        return self.d.an.ad

    def frame(self, src_super, ref_super, old_blob):
        blob = np.zeros(self.blob_size, dtype=np.uint8)
        sp, spitch = _planes(src_super)
        ob = np.ascontiguousarray(old_blob)
        if ref_super is None:
            lib().mvo_recalculate_frame(C.byref(self.d), sp, spitch, None, None, C.c_void_p(ob.ctypes.data), C.c_void_p(blob.ctypes.data))
        else:
            rp, rpitch = _planes(ref_super)
            lib().mvo_recalculate_frame(C.byref(self.d), sp, spitch, rp, rpitch, C.c_void_p(ob.ctypes.data), C.c_void_p(blob.ctypes.data))
        return blob
