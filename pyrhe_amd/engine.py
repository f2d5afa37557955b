"""ctypes binding of the C ABI in ``include/pyrhe_sketch.h`` + a thin torch-tensor wrapper.

PyTorch is plumbing here (device memory, streams); every byte of the sketch is computed by the
HIP kernels in ``csrc/`` (``sketch_engine.hip`` + ``kernels_*.inc``).  There is NO CPU fallback: if the shared library is
missing or no GPU is present, construction raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import numpy as np
import torch

from .build import library_path

PREC_F16 = 0
PREC_F16X2 = 1
PREC_F16_EXACT = 2
PREC_I8X3 = 3
PRECISIONS = {"f16": PREC_F16, "f16x2": PREC_F16X2, "f16e": PREC_F16_EXACT, "i8x3": PREC_I8X3, "i8f16": 4}
COL_SKETCH = 1
COL_NORM = 2
COL_CROSS = 3
ROW_ALIGN = 256           # pitch granularity (bytes)

_lib = None

# name -> (restype, argtypes): exactly the symbols include/pyrhe_sketch.h declares
_vp, _i32, _i64, _u64 = C.c_void_p, C.c_int, C.c_int64, C.c_uint64
SYMBOLS = {
    "pyrhe_last_error": (C.c_char_p, []),
    "pyrhe_abi_version": (_i32, []),
    "pyrhe_engine_create": (_i32, [_i32, _i64, _i64, _i64, _i32, _i64, C.POINTER(_vp)]),
    "pyrhe_engine_destroy": (_i32, [_vp]),
    "pyrhe_engine_set_rhs": (_i32, [_vp, _vp, C.POINTER(C.c_int32), _i32, _vp, _vp]),
    "pyrhe_engine_sketch_block": (_i32, [_vp, _vp, _i64, _vp, C.POINTER(_i64), _i32, _vp, _vp, _vp, _vp, _i32, _vp]),
    "pyrhe_engine_sketch_block_dom": (_i32, [_vp, _vp, _i64, _vp, C.POINTER(_i64), _i32, _vp, _vp, _vp, _vp, _vp, _i32, _vp]),
    "pyrhe_engine_sketch_block_cross": (_i32, [_vp, _vp, _i64, _vp, C.POINTER(_i64), _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp]),
    "pyrhe_engine_block_stats": (_i32, [_vp, _vp, _i64, _vp, _vp, _vp]),
    "pyrhe_engine_set_async": (_i32, [_vp, _i32]),
    "pyrhe_engine_join": (_i32, [_vp, _vp]),
    "pyrhe_engine_set_pipeline": (_i32, [_vp, _i32]),
    "pyrhe_engine_set_option": (_i32, [_vp, _i32, _i64]),
    "pyrhe_decode_rows": (_i32, [_vp, _i64, _i64, _vp, _vp]),
    "pyrhe_recode_rows": (_i32, [_vp, _i64, _i64, _i32, _vp]),
    "pyrhe_engine_set_coding": (_i32, [_vp, _i32]),
    "pyrhe_sum_blocks": (_i32, [_vp, _i32, _i64, _i64, _vp, _vp]),
    "pyrhe_sum_blocks_weighted": (_i32, [_vp, _i32, _i64, _i64, _vp, _vp, _vp]),
    "pyrhe_sum_blocks_both": (_i32, [_vp, _i32, _i64, _i64, _vp, _vp, _vp, _vp]),
    "pyrhe_engine_loo_dots": (_i32, [_vp, _vp, _i32, _vp, _i32, _i64, _i32, _i32, _vp, _vp]),
    "pyrhe_engine_loo_dots_multi": (_i32, [_vp, _vp, _i32, _vp, _i32, _vp, _i32, _i32, _i64, _i32, _i32, _vp, _vp, _vp, _vp]),
    "pyrhe_engine_apply_projector": (_i32, [_vp, _vp, _i64, _i64, _vp, _vp, _i32, _vp, _vp]),
    "pyrhe_synth_genotypes": (_i32, [_vp, _i64, _i64, _i64, _u64, _i64, _vp]),
    "pyrhe_stream_read": (_i32, [_vp, _i64, _vp, _vp]),
    "pyrhe_engine_info": (_i64, [_vp, _i32]),
    "pyrhe_engine_profile": (_i32, [_vp, _i32]),
    "pyrhe_engine_profile_read": (_i32, [_vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(_i64)]),
}


def load_library():
    """dlopen ``libpyrhe_sketch.so`` (built in-tree by ``pyrhe_amd.build``) and bind every symbol."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise RuntimeError(f"{path} not found: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(the sketch engine has no CPU fallback)")
    lib = C.CDLL(path)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)      # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


class SketchError(RuntimeError):
    pass


class _RowStruct(C.Structure):      # pyrhe_row_t (include/pyrhe_sketch.h)
    _fields_ = [("tot", _vp), ("blk", _vp), ("mul", _vp), ("blk_batch_stride", _i64)]


class Row:
    """One logical row of ``pyrhe_engine_loo_dots``: ``tot - blk[z] * mul``.  ``tot`` float64 ``[n_sub * period]``
    view, ``blk`` float32 view of batch element 0 with ``stride`` elements to the next one, ``mul`` float32 ``[period]``."""
    __slots__ = ("tot", "blk", "mul", "stride")

    def __init__(self, tot=None, blk=None, mul=None, stride=0):
        for t, dt in ((tot, torch.float64), (blk, torch.float32), (mul, torch.float32)):
            assert t is None or (t.dtype == dt and t.is_contiguous())
        self.tot, self.blk, self.mul, self.stride = tot, blk, mul, int(stride)


def _check(code: int, what: str):
    if code != 0:
        msg = load_library().pyrhe_last_error().decode(errors="replace")
        raise SketchError(f"{what} failed (code {code}): {msg}")


def pitch_for(n_samples_orig: int) -> int:
    """Row pitch in bytes: ceil(N/4) rounded up to 256."""
    rb = (n_samples_orig + 3) // 4
    return (rb + ROW_ALIGN - 1) // ROW_ALIGN * ROW_ALIGN


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream_ptr(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _require_gpu(device) -> torch.device:
    if not torch.cuda.is_available():
        raise RuntimeError("pyrhe_amd: no ROCm GPU visible -- the sketch engine has no CPU fallback")
    device = torch.device(device)
    if device.type != "cuda":
        raise RuntimeError(f"pyrhe_amd: device must be a ROCm GPU, got {device}")
    if device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    return device


_STAGING = {}        # "pieces": the pinned staging pieces kept between uploads (one set; re-made when a call needs wider or more rows)


def _staging(rows_per: int, sw: int, n_stage: int):
    """Pinned staging pieces, kept between calls: page-locking 2 x 256 MiB costs 0.1 s -- nothing once per cohort, a third of the
    transfer time when a host-resident shard is uploaded window by window (HostRows).  One set at a time."""
    have = _STAGING.get("pieces")
    if have is None or have[0].shape[1] != sw or have[0].shape[0] < rows_per or len(have) < n_stage:
        _STAGING.clear()
        _STAGING["pieces"] = have = [torch.zeros((rows_per, sw), dtype=torch.uint8).pin_memory() for _ in range(n_stage)]
    return [t[:rows_per] for t in have[:n_stage]]          # (a smaller request -- the last window of a shard -- takes views)


def _file_backing(a: np.ndarray):
    """``(path, byte offset of a[0, 0])`` when ``a`` is a C-contiguous view into a memory-mapped file (np.memmap), else None."""
    if not isinstance(a, np.ndarray) or not a.flags.c_contiguous:
        return None
    root = a
    while isinstance(getattr(root, "base", None), np.ndarray):
        root = root.base
    if not isinstance(root, np.memmap) or not getattr(root, "filename", None):
        return None
    off = int(root.offset) + (a.__array_interface__["data"][0] - root.__array_interface__["data"][0])
    return str(root.filename), off


def upload_packed(packed_rows: np.ndarray, device, chunk_bytes: int = 1 << 28, recode: Optional[str] = None,
                  threads: Optional[int] = None, n_stage: int = 2, pitched: bool = False) -> torch.Tensor:
    """Host ``[m, ceil(N/4)]`` uint8 rows (a NumPy array or a memory-mapped ``.bed`` body) -> device ``[m, pitch]``,
    zero padded.  The rows are never decoded on the host; they go through ``n_stage`` pinned staging pieces of ``chunk_bytes``
    (the host-side fill of piece i+1 overlaps the transfer of piece i), so a memory-mapped 100-GB file needs neither
    100 GB of RAM nor one giant copy.  Measured on the MI355X box (tools/upload_rate.py, 8 GB): with the staging
    pieces KEPT between calls (_staging: page-locking them anew on every call was 0.1 s of a 0.33 s upload) the link takes
    48-54 GB/s from an array in RAM -- 24-25 before --, filled by ``threads`` workers (default 4; NumPy releases the GIL in the
    copy).  A memory-mapped FILE read through page faults fed it at 6-8 GB/s: rows that are a view of a mapped file are
    therefore read with ``os.preadv`` straight into the pinned piece, ``threads`` slices at a time: 36-42 GB/s (page cache warm).
    An upload of a few hundred MB (one window of a host-resident shard) goes in about eight pieces, so that fills and transfers
    overlap inside it as well.
    ``pitched``: staging rows carry the device pitch (one contiguous copy per piece instead of a 2-D copy; arrays only).
    ``recode="value"``: every piece is recoded in place to the 2-bit value code (``pyrhe_recode_rows``) on a second stream as
    soon as it has landed, WHILE the next piece crosses PCIe -- the one-time transcode (51 ms per 125 GB of kernel time) then
    costs the load nothing but the last piece's 0.1 ms, where a recode after the upload cost a single-trait run (one sketch
    pass of 46 ms) more than the 2 ms per pass it saves."""
    from concurrent.futures import ThreadPoolExecutor
    device = _require_gpu(device)
    m, rb = packed_rows.shape
    pitch = (rb + ROW_ALIGN - 1) // ROW_ALIGN * ROW_ALIGN
    if m == 0:
        return torch.zeros((0, pitch), dtype=torch.uint8, device=device)
    backing = _file_backing(packed_rows)
    if backing is not None:
        pitched = False                                             # file bytes land as they lie: rows of rb bytes
    sw = pitch if pitched else rb                                   # bytes per staged row
    out = torch.empty((m, pitch), dtype=torch.uint8, device=device) if pitched else torch.zeros((m, pitch), dtype=torch.uint8, device=device)
    # a small upload (one window of a host-resident shard: a few hundred MB) goes in about eight pieces, so that the fill of a
    # piece overlaps the transfer of the one before it inside the window too
    chunk_bytes = max(min(chunk_bytes, 32 << 20), min(chunk_bytes, (m * sw + 7) // 8))
    rows_per = max(1, min(chunk_bytes // max(sw, 1), m))
    n_pieces = -(-m // rows_per)
    n_stage = max(1, min(int(n_stage), n_pieces))
    staging = _staging(rows_per, sw, n_stage)
    if threads is None:
        try:
            avail = len(os.sched_getaffinity(0))
        except AttributeError:
            avail = os.cpu_count() or 1
        threads = max(1, min(4, avail))
    threads = max(1, min(int(threads), rows_per))
    pool = ThreadPoolExecutor(threads) if threads > 1 else None
    done = [None] * n_stage
    fd = os.open(backing[0], os.O_RDONLY) if backing is not None else -1

    def fill_file(dst, r0, r1):
        """rows r0 .. r1 - 1 of the file -> the first (r1 - r0) * rb bytes of the piece, by positional reads"""
        flat = dst.reshape(-1)
        nbytes, base = (r1 - r0) * rb, backing[1] + r0 * rb

        def read(a, b):
            pos = a
            while pos < b:
                got = os.preadv(fd, [memoryview(flat[pos:b])], base + pos)
                if got <= 0:
                    raise IOError(f"{backing[0]}: short read at byte {base + pos}")
                pos += got
        step = -(-nbytes // threads) if pool is not None else nbytes
        step = max(1 << 20, (step + 4095) // 4096 * 4096)
        spans = [(a, min(a + step, nbytes)) for a in range(0, nbytes, step)]
        if pool is None or len(spans) == 1:
            for a, b in spans:
                read(a, b)
        else:
            for j in [pool.submit(read, a, b) for a, b in spans]:
                j.result()

    def fill(dst, r0, r1):
        """rows r0 .. r1 - 1 of the source -> the first r1 - r0 staged rows"""
        if backing is not None:
            return fill_file(dst, r0, r1)
        n = r1 - r0
        if pool is None or n < 4 * threads:
            dst[:n, :rb] = packed_rows[r0:r1]
            return
        step = -(-n // threads)
        jobs = [pool.submit(lambda a=a: dst.__setitem__((slice(a, min(a + step, n)), slice(0, rb)), packed_rows[r0 + a:r0 + min(a + step, n)]))
                for a in range(0, n, step)]
        for j in jobs:
            j.result()
    try:
        with torch.cuda.device(device):
            side = torch.cuda.Stream(device) if recode else None
            for i, r0 in enumerate(range(0, m, rows_per)):
                r1 = min(r0 + rows_per, m)
                buf = staging[i % n_stage]
                if done[i % n_stage] is not None:
                    done[i % n_stage].synchronize()               # the transfer that last used this piece
                fill(buf.numpy(), r0, r1)
                if pitched:
                    out[r0:r1].copy_(buf[: r1 - r0], non_blocking=True)
                else:
                    out[r0:r1, :rb].copy_(buf[: r1 - r0], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
                done[i % n_stage] = ev
                if recode:
                    side.wait_event(ev)                           # the piece has landed (and `out` was zeroed before it)
                    with torch.cuda.stream(side):
                        recode_rows(out[r0:r1], recode)
            torch.cuda.current_stream(device).synchronize()       # (also: the cached pieces are free for the next call)
            if recode:
                side.synchronize()
    finally:
        if pool is not None:
            pool.shutdown(wait=True)
        if fd >= 0:
            os.close(fd)
    return out


class HostRows:
    """Genotype rows that STAY ON THE HOST (a NumPy array or the memory-mapped body of a ``.bed`` file) and reach the device a
    window at a time -- for a shard that does not fit HBM beside the sketch store (N = 500k: more than ~2 M SNPs per GPU).
    ``window(a, b)`` returns rows a .. b - 1 as a device ``[b - a, pitch]`` tensor in the coding ``recode`` names (uploaded
    through :func:`upload_packed`: pinned staging, threaded host copies, the transcode overlapped); the most recent window is
    kept, so a repeated or nested request costs nothing.  SketchRunner takes it in place of the resident device tensor: every
    engine call then sketches one window (its blocks' rows) -- the per-SNP statistics behind the imputation are taken from
    the same window right before its sketch, not in a pass over the link of their own --, pass 2 works on the stored sketches
    and never asks for rows again (unless it re-sketches: ``recompute_pass2``).  The pass is then bound by the PCIe link
    (36-53 GB/s = 0.3-0.4 M SNPs/s at N = 500k), not by HBM: a mode for cohorts that would otherwise not run at all, not a
    fast path."""

    def __init__(self, host_rows: np.ndarray, device, recode: Optional[str] = "value"):
        assert host_rows.ndim == 2 and host_rows.dtype == np.uint8
        self.host = host_rows
        self.device = _require_gpu(device)
        self.recode = recode
        m, rb = host_rows.shape
        self.pitch = (rb + ROW_ALIGN - 1) // ROW_ALIGN * ROW_ALIGN
        self.shape = (m, self.pitch)
        self.dtype = torch.uint8
        self._win = None
        self.uploads = 0            # windows uploaded so far (tests, logs)
        self.upload_bytes = 0

    def window(self, a: int, b: int) -> torch.Tensor:
        a, b = int(a), int(b)
        assert 0 <= a <= b <= self.shape[0]
        if self._win is not None and self._win[0] <= a and b <= self._win[1]:
            return self._win[2][a - self._win[0]: b - self._win[0]]
        self._win = None            # release the old window before the new one is allocated
        t = upload_packed(self.host[a:b], self.device, recode=self.recode)
        self._win = (a, b, t)
        self.uploads += 1
        self.upload_bytes += int(t.numel())
        return t

    def release(self):
        self._win = None


CODINGS = {"plink": 0, "value": 1}
# PYRHE_OPT_* of include/pyrhe_sketch.h: launch-shape knobs, every value computes the same results
OPTIONS = {"g1_workgroups": 1, "g2_packed": 2, "g2_iw": 3, "g2_splits": 4, "dots_generic": 5, "g1_split_major": 6, "g1_wide": 7, "g2_merge": 8, "g1_zero_rows": 9, "list_span": 10, "g1_fill": 11, "dom_seg": 12, "dots_f32": 13, "dots_tiles": 14, "g1_line": 15, "dom_fuse": 16, "g1_compact": 17,
           "diag_g1_plain_store": 100, "diag_g1_no_barrier": 101, "diag_g1_stats": 102}     # 100+: only in the diagnostic build (PYRHE_LIB=...libpyrhe_sketch_diag.so), wrong results


def recode_rows(packed: torch.Tensor, to: str = "value") -> torch.Tensor:
    """In-place recode of device rows between the PLINK code and the 2-bit value code (``pyrhe_recode_rows``); tell the
    engines that will read the buffer with :meth:`SketchEngine.set_coding`."""
    assert packed.dtype == torch.uint8 and packed.is_contiguous() and packed.dim() == 2
    _require_gpu(packed.device)
    with torch.cuda.device(packed.device):
        _check(load_library().pyrhe_recode_rows(_ptr(packed), packed.shape[0], packed.shape[1], CODINGS[to],
                                                _stream_ptr(packed.device)), "pyrhe_recode_rows")
    return packed


def stream_read_gbs(buf: torch.Tensor, repeats: int = 3) -> float:
    """Measured plain read-stream bandwidth (GB/s) over ``buf`` (``pyrhe_stream_read``; best of ``repeats``, HIP events on the
    current stream)."""
    lib = load_library()
    dev = _require_gpu(buf.device)
    nbytes = buf.numel() * buf.element_size() // 16 * 16
    sink = torch.zeros(1, dtype=torch.int32, device=dev)
    best = 0.0
    with torch.cuda.device(dev):
        for _ in range(repeats + 1):                      # the first pass is a warm-up
            t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0.record()
            _check(lib.pyrhe_stream_read(_ptr(buf), nbytes, _ptr(sink), _stream_ptr(dev)), "pyrhe_stream_read")
            t1.record()
            t1.synchronize()
            if _ > 0:
                best = max(best, nbytes / (t0.elapsed_time(t1) * 1e-3) / 1e9)
    return best


def synth_genotypes(n_rows: int, n_samples: int, seed: int, row0: int, device, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Device-side synthetic cohort rows (see ``pyrhe_synth_genotypes``)."""
    device = _require_gpu(device)
    lib = load_library()
    pitch = pitch_for(n_samples)
    if out is None:
        out = torch.empty((n_rows, pitch), dtype=torch.uint8, device=device)
    assert out.shape == (n_rows, pitch) and out.is_contiguous()
    step = max(1, (1 << 30) // pitch)          # keep each launch's grid well inside 2^31 blocks
    with torch.cuda.device(device):
        for r in range(0, n_rows, step):
            n = min(step, n_rows - r)
            _check(lib.pyrhe_synth_genotypes(C.c_void_p(out[r].data_ptr()), n, pitch, n_samples, seed & (2**64 - 1), row0 + r,
                                             _stream_ptr(device)), "pyrhe_synth_genotypes")
    return out


def sum_blocks(blocks: torch.Tensor, weights: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``blocks [jl, ...]`` float32 (contiguous, on the GPU) -> float64 (weighted) sum over the first dimension by
    ``pyrhe_sum_blocks[_weighted]``.  ``weights``: float64 ``[jl]`` on the same device."""
    _require_gpu(blocks.device)
    assert blocks.dtype == torch.float32 and blocks.is_contiguous()
    out = torch.zeros(blocks.shape[1:], dtype=torch.float64, device=blocks.device)
    if blocks.shape[0] == 0:
        return out
    n = blocks[0].numel()
    if n % 4:
        raise ValueError("sum_blocks: the block size must be a multiple of 4 elements")
    lib = load_library()
    with torch.cuda.device(blocks.device):
        if weights is None:
            _check(lib.pyrhe_sum_blocks(_ptr(blocks), blocks.shape[0], n, n, _ptr(out), _stream_ptr(blocks.device)),
                   "pyrhe_sum_blocks")
        else:
            assert weights.dtype == torch.float64 and weights.is_contiguous() and weights.numel() == blocks.shape[0]
            _check(lib.pyrhe_sum_blocks_weighted(_ptr(blocks), blocks.shape[0], n, n, _ptr(weights), _ptr(out),
                                                 _stream_ptr(blocks.device)), "pyrhe_sum_blocks_weighted")
    return out


def sum_blocks_both(blocks: torch.Tensor, weights: torch.Tensor):
    """The plain and the weighted float64 sum over the first dimension of ``blocks [jl, ...]`` from ONE read of them
    (``pyrhe_sum_blocks_both``); returns ``(sum, weighted sum)``, the same bits as the two :func:`sum_blocks` calls."""
    _require_gpu(blocks.device)
    assert blocks.dtype == torch.float32 and blocks.is_contiguous()
    assert weights.dtype == torch.float64 and weights.is_contiguous() and weights.numel() == blocks.shape[0]
    out = torch.zeros(blocks.shape[1:], dtype=torch.float64, device=blocks.device)
    out_w = torch.zeros_like(out)
    if blocks.shape[0] == 0:
        return out, out_w
    n = blocks[0].numel()
    if n % 4:
        raise ValueError("sum_blocks_both: the block size must be a multiple of 4 elements")
    with torch.cuda.device(blocks.device):
        _check(load_library().pyrhe_sum_blocks_both(_ptr(blocks), blocks.shape[0], n, n, _ptr(weights), _ptr(out), _ptr(out_w),
                                                    _stream_ptr(blocks.device)), "pyrhe_sum_blocks_both")
    return out, out_w


def decode_rows(packed: torch.Tensor) -> torch.Tensor:
    """Dense float32 ``[m, 4*pitch]`` decode of packed rows (NaN = missing call)."""
    lib = load_library()
    m, pitch = packed.shape
    out = torch.empty((m, 4 * pitch), dtype=torch.float32, device=packed.device)
    with torch.cuda.device(packed.device):
        _check(lib.pyrhe_decode_rows(_ptr(packed), m, pitch, _ptr(out), _stream_ptr(packed.device)), "pyrhe_decode_rows")
    return out


class SketchEngine:
    """One engine per GPU.  Mirrors what ``Base._compute_XXz/_XXUz/_yXXy`` + ``standardize_geno`` +
    ``read_geno`` do for a block (reference ``pyrhe/src/base/base.py:291-296,338-356,403-417``)."""

    def __init__(self, n_samples_orig: int, n_keep: int, max_entries: int, precision: str = "i8x3",
                 device="cuda", sub_snps: int = 0):
        self.device = _require_gpu(device)
        self.lib = load_library()
        if precision not in PRECISIONS:
            raise ValueError(f"precision must be one of {list(PRECISIONS)}")
        self.precision = precision
        self.n_samples_orig = int(n_samples_orig)
        self.n_keep = int(n_keep)
        self.pitch = pitch_for(n_samples_orig)
        self.n_pad = 4 * self.pitch
        self.max_entries = int(max_entries)
        self.coding = "plink"
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            _check(self.lib.pyrhe_engine_create(self.device.index, self.n_keep, self.pitch, self.max_entries,
                                                PRECISIONS[precision], int(sub_snps), C.byref(h)), "pyrhe_engine_create")
        self._h = h
        self.n_cu = int(torch.cuda.get_device_properties(self.device).multi_processor_count)
        self.n_sketch = 0
        self.options = {}             # knobs set through set_option (the library's defaults otherwise)
        self.n_norm = 0
        self.n_cross = 0
        self._keep = None
        self._rhs = None

    def close(self):
        if getattr(self, "_h", None):
            self.lib.pyrhe_engine_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def sub_snps(self) -> int:
        return int(self.lib.pyrhe_engine_info(self._h, 1))

    def keep_vector(self, missing_indv: Sequence[int] = ()) -> torch.Tensor:
        keep = torch.zeros(self.n_pad, dtype=torch.float32, device=self.device)
        keep[: self.n_samples_orig] = 1.0
        if len(missing_indv):
            keep[torch.as_tensor(list(missing_indv), device=self.device, dtype=torch.long)] = 0.0
        return keep

    def set_rhs(self, sketch_cols: Optional[torch.Tensor], norm_cols: Optional[torch.Tensor], keep: torch.Tensor,
                cross_cols: Optional[torch.Tensor] = None, cross_digits: int = 3, norm_digits: int = 3):
        """sketch_cols ``[n_sketch, n_pad]`` / norm_cols ``[n_norm, n_pad]`` / cross_cols ``[n_cross, n_pad]`` float32 on the
        device, zero at dropped and padding samples.  ``cross_digits`` (1..3): int8 digits the cross columns are fed with
        (``PYRHE_COL_DIGITS``) -- one count for all of them, or one per cross column (not growing with the column index)."""
        parts = [t for t in (sketch_cols, norm_cols, cross_cols) if t is not None and t.shape[0] > 0]
        R = torch.cat(parts, dim=0).to(torch.float32).contiguous()
        assert R.shape[1] == self.n_pad and R.device == self.device
        ns = 0 if sketch_cols is None else sketch_cols.shape[0]
        nn = 0 if norm_cols is None else norm_cols.shape[0]
        nx = 0 if cross_cols is None else cross_cols.shape[0]
        per_col = [int(d) for d in cross_digits] if isinstance(cross_digits, (list, tuple, np.ndarray)) else [int(cross_digits)] * nx
        assert len(per_col) == nx and all(1 <= d <= 3 for d in per_col) and 1 <= int(norm_digits) <= 3
        per_col = [min(d, int(norm_digits)) for d in per_col]           # digit counts do not grow with the column index
        roles = (C.c_int32 * (ns + nn + nx))(*([COL_SKETCH] * ns + [COL_NORM | (int(norm_digits) << 8)] * nn +
                                               [COL_CROSS | (d << 8) for d in per_col]))
        keep = keep.to(torch.float32).contiguous()
        with torch.cuda.device(self.device):
            _check(self.lib.pyrhe_engine_set_rhs(self._h, _ptr(R), roles, ns + nn + nx, _ptr(keep), _stream_ptr(self.device)),
                   "pyrhe_engine_set_rhs")
        self.n_sketch, self.n_norm, self.n_cross = ns, nn, nx
        self._rhs, self._keep = R, keep      # keep alive until the async prep has run

    def sketch_block(self, packed: torch.Tensor, entries: torch.Tensor, offsets: Sequence[int], *,
                     fill: Optional[torch.Tensor] = None, out_sketch: Optional[torch.Tensor] = None,
                     out_norm: Optional[torch.Tensor] = None, accumulate: bool = False, want_colsum: bool = False,
                     out_cross: Optional[torch.Tensor] = None):
        """Enqueue the sketch of one block; returns ``(out_sketch [nbin, n_sketch, n_pad], out_norm [nbin, n_norm], colsum|None)``.
        With cross columns installed ``out_cross [nbin, n_cross, n_cross]`` float64 receives their Gram matrices
        (``pyrhe_engine_sketch_block_cross``)."""
        nbin = len(offsets) - 1
        assert packed.dtype == torch.uint8 and packed.is_contiguous() and packed.shape[1] == self.pitch
        assert entries.dtype == torch.int32 and entries.is_contiguous()
        if out_sketch is None:
            out_sketch = torch.empty((nbin, self.n_sketch, self.n_pad), dtype=torch.float32, device=self.device)
            assert not accumulate
        if out_norm is None:
            out_norm = torch.empty((nbin, max(self.n_norm, 1)), dtype=torch.float64, device=self.device)
            assert not accumulate
        colsum = torch.empty(int(offsets[-1]), dtype=torch.int32, device=self.device) if want_colsum else None
        offs = (C.c_int64 * (nbin + 1))(*[int(o) for o in offsets])
        if self.n_cross:
            assert not want_colsum and out_cross is not None and out_cross.dtype == torch.float64 and out_cross.is_contiguous()
            assert out_cross.numel() == nbin * self.n_cross * self.n_cross
            with torch.cuda.device(self.device):
                _check(self.lib.pyrhe_engine_sketch_block_cross(self._h, _ptr(packed), packed.shape[0], _ptr(entries), offs, nbin,
                                                                _ptr(fill), _ptr(out_sketch), _ptr(out_norm), _ptr(out_cross),
                                                                None, None, None, 1 if accumulate else 0,
                                                                _stream_ptr(self.device)), "pyrhe_engine_sketch_block_cross")
            return out_sketch, out_norm, None
        with torch.cuda.device(self.device):
            _check(self.lib.pyrhe_engine_sketch_block(self._h, _ptr(packed), packed.shape[0], _ptr(entries), offs, nbin,
                                                      _ptr(fill), _ptr(out_sketch), _ptr(out_norm), _ptr(colsum),
                                                      1 if accumulate else 0, _stream_ptr(self.device)),
                   "pyrhe_engine_sketch_block")
        return out_sketch, out_norm, colsum

    def sketch_block_dom(self, packed: torch.Tensor, entries: torch.Tensor, offsets: Sequence[int], *,
                         fill: Optional[torch.Tensor] = None, out_sketch=None, out_norm=None, out_sketch_dom=None,
                         out_norm_dom=None, accumulate: bool = False, out_cross=None, out_cross_dom=None):
        """Additive + dominance (RHE_DOM) sketches of one block; returns the four output tensors (with cross columns
        installed their Gram matrices under both encodings go to ``out_cross`` / ``out_cross_dom``)."""
        nbin = len(offsets) - 1
        assert packed.dtype == torch.uint8 and packed.is_contiguous() and packed.shape[1] == self.pitch

        def _mk(t, shape, dtype):
            if t is None:
                assert not accumulate
                return torch.empty(shape, dtype=dtype, device=self.device)
            return t
        out_sketch = _mk(out_sketch, (nbin, self.n_sketch, self.n_pad), torch.float32)
        out_sketch_dom = _mk(out_sketch_dom, (nbin, self.n_sketch, self.n_pad), torch.float32)
        out_norm = _mk(out_norm, (nbin, max(self.n_norm, 1)), torch.float64)
        out_norm_dom = _mk(out_norm_dom, (nbin, max(self.n_norm, 1)), torch.float64)
        offs = (C.c_int64 * (nbin + 1))(*[int(o) for o in offsets])
        if self.n_cross:
            assert out_cross is not None and out_cross_dom is not None
            with torch.cuda.device(self.device):
                _check(self.lib.pyrhe_engine_sketch_block_cross(self._h, _ptr(packed), packed.shape[0], _ptr(entries), offs, nbin,
                                                                _ptr(fill), _ptr(out_sketch), _ptr(out_norm), _ptr(out_cross),
                                                                _ptr(out_sketch_dom), _ptr(out_norm_dom), _ptr(out_cross_dom),
                                                                1 if accumulate else 0, _stream_ptr(self.device)),
                       "pyrhe_engine_sketch_block_cross")
            return out_sketch, out_norm, out_sketch_dom, out_norm_dom
        with torch.cuda.device(self.device):
            _check(self.lib.pyrhe_engine_sketch_block_dom(self._h, _ptr(packed), packed.shape[0], _ptr(entries), offs, nbin,
                                                          _ptr(fill), _ptr(out_sketch), _ptr(out_norm), _ptr(out_sketch_dom),
                                                          _ptr(out_norm_dom), 1 if accumulate else 0, _stream_ptr(self.device)),
                   "pyrhe_engine_sketch_block_dom")
        return out_sketch, out_norm, out_sketch_dom, out_norm_dom

    def sum_blocks_both(self, blocks: torch.Tensor, weights: torch.Tensor):
        """(plain, weighted) float64 sums over the block dimension from one read of the store."""
        return sum_blocks_both(blocks, weights)

    def sum_blocks(self, blocks: torch.Tensor, weights: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Float64 (weighted) sum over the stored per-block sketches (the reference's ``aggregate``)."""
        return sum_blocks(blocks, weights)

    def set_coding(self, coding: str):
        """How the ``packed`` buffers handed to this engine are coded: ``"plink"`` (default) or ``"value"``."""
        _check(self.lib.pyrhe_engine_set_coding(self._h, CODINGS[coding]), "pyrhe_engine_set_coding")
        self.coding = coding

    def set_option(self, name: str, value: int):
        """Launch-shape knob of ``pyrhe_engine_set_option`` (see ``OPTIONS``)."""
        _check(self.lib.pyrhe_engine_set_option(self._h, OPTIONS[name], int(value)), f"pyrhe_engine_set_option({name})")
        self.options[name] = int(value)

    def last_path(self) -> dict:
        """Which kernels the last sketch call launched (``pyrhe_engine_info`` 6..8): ``gemm1`` = 100 + tiles for the int8
        kernel / tiles for the fp16 kernel, ``gemm2`` = 11 / 12 packed int8 kernel with one / two column groups, 1 fp16
        kernel, 0 none; ``segmented`` = one GEMM 1 over all lists of the call; ``g2_merged`` = segments per workgroup of the
        last packed GEMM 2 launch (> 1: short lists shared workgroups, ``g2_merge`` option)."""
        g1, g2, fl, mg = (int(self.lib.pyrhe_engine_info(self._h, w)) for w in (6, 7, 8, 9))
        return {"gemm1": g1, "gemm2": g2, "segmented": bool(fl), "g2_merged": mg}

    def set_async(self, enable: bool):
        """Defer the per-call join: consecutive blocks then overlap (GEMM 1 of block j+1 beside GEMM 2 of block j)."""
        _check(self.lib.pyrhe_engine_set_async(self._h, 1 if enable else 0), "pyrhe_engine_set_async")

    def set_pipeline(self, enable: bool):
        _check(self.lib.pyrhe_engine_set_pipeline(self._h, 1 if enable else 0), "pyrhe_engine_set_pipeline")

    def join(self):
        """Order every un-joined sketch into the current torch stream."""
        with torch.cuda.device(self.device):
            _check(self.lib.pyrhe_engine_join(self._h, _stream_ptr(self.device)), "pyrhe_engine_join")

    def block_stats(self, packed: torch.Tensor):
        m = packed.shape[0]
        s = torch.empty(m, dtype=torch.int32, device=self.device)
        nm = torch.empty(m, dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            _check(self.lib.pyrhe_engine_block_stats(self._h, _ptr(packed), m, _ptr(s), _ptr(nm), _stream_ptr(self.device)),
                   "pyrhe_engine_block_stats")
        return s, nm

    def loo_dots(self, x: Sequence[Row], y: Sequence[Row], period: int, n_sub: int, n_batch: int = 1) -> torch.Tensor:
        """``out[z, a, b] = sum_idx x_a[idx] y_b[idx]`` over ``n_sub * period`` elements, float64 ``[n_batch, nx, ny]``."""
        def pack(rows):
            arr = (_RowStruct * len(rows))()
            for i, r in enumerate(rows):
                arr[i].tot = None if r.tot is None else r.tot.data_ptr()
                arr[i].blk = None if r.blk is None else r.blk.data_ptr()
                arr[i].mul = None if r.mul is None else r.mul.data_ptr()
                arr[i].blk_batch_stride = r.stride
            return arr
        out = torch.empty((n_batch, len(x), len(y)), dtype=torch.float64, device=self.device)
        xa, ya = pack(x), pack(y)
        with torch.cuda.device(self.device):
            _check(self.lib.pyrhe_engine_loo_dots(self._h, C.cast(xa, _vp), len(x), C.cast(ya, _vp), len(y), int(period),
                                                  int(n_sub), int(n_batch), _ptr(out), _stream_ptr(self.device)),
                   "pyrhe_engine_loo_dots")
        return out

    def apply_projector(self, x: torch.Tensor, Wt32: torch.Tensor, Q32: torch.Tensor) -> torch.Tensor:
        """``W (Q (W^T x))`` for every row of ``x [..., n]`` float64 (``pyrhe_engine_apply_projector``: the reference's
        float32 ``_compute_UXXz``, base.py:407-408).  ``Wt32 [c, n]`` / ``Q32 [c, c]`` float32 on the device; float64 out."""
        assert x.dtype == torch.float64 and x.is_contiguous() and Wt32.dtype == torch.float32 and Wt32.is_contiguous()
        assert Q32.dtype == torch.float32 and Q32.is_contiguous() and Wt32.shape[1] == x.shape[-1]
        out = torch.empty_like(x)
        n = x.shape[-1]
        n_rows = x.numel() // n if n else 0
        with torch.cuda.device(self.device):
            _check(self.lib.pyrhe_engine_apply_projector(self._h, _ptr(x), n_rows, n, _ptr(Wt32), _ptr(Q32), Wt32.shape[0],
                                                         _ptr(out), _stream_ptr(self.device)), "pyrhe_engine_apply_projector")
        return out

    @staticmethod
    def multi_ok(x: Sequence[Row], y_sub: Sequence[Row], y_flat: Sequence[Row], period: int) -> bool:
        """Shapes ``loo_dots_multi`` takes: 1..16 aligned block-only rows, 1..16 / 0..16 tot-only fixed rows."""
        def blk_ok(r):
            return (r.tot is None and r.blk is not None and r.blk.data_ptr() % 16 == 0 and r.stride % 4 == 0
                    and (r.mul is None or r.mul.data_ptr() % 16 == 0))
        def tot_ok(r):
            return r.tot is not None and r.blk is None and r.tot.data_ptr() % 16 == 0
        return (1 <= len(x) <= 16 and 1 <= len(y_sub) <= 16 and len(y_flat) <= 16 and period % 32 == 0
                and all(blk_ok(r) for r in x) and all(tot_ok(r) for r in y_sub) and all(tot_ok(r) for r in y_flat))

    def loo_dots_multi(self, x: Sequence[Row], y_sub: Sequence[Row], y_flat: Sequence[Row], want_self: bool, period: int,
                       n_sub: int, n_batch: int):
        """One pass over the block rows ``x`` (evaluated as ``-blk * mul``) for ``out_sub [n_batch, nx, ny_sub]`` (summed over
        the sub-rows), ``out_flat [n_batch, nx, n_sub, ny_flat]`` (one result per sub-row; None without flat rows) and
        ``out_self [n_batch, nx, nx]`` (None unless ``want_self``): see ``pyrhe_engine_loo_dots_multi``."""
        def pack(rows):
            arr = (_RowStruct * max(len(rows), 1))()
            for i, r in enumerate(rows):
                arr[i].tot = None if r.tot is None else r.tot.data_ptr()
                arr[i].blk = None if r.blk is None else r.blk.data_ptr()
                arr[i].mul = None if r.mul is None else r.mul.data_ptr()
                arr[i].blk_batch_stride = r.stride
            return arr
        f64 = dict(dtype=torch.float64, device=self.device)
        out_sub = torch.empty((n_batch, len(x), len(y_sub)), **f64)
        out_flat = torch.empty((n_batch, len(x), n_sub, len(y_flat)), **f64) if y_flat else None
        out_self = torch.empty((n_batch, len(x), len(x)), **f64) if want_self else None
        xa, sa, fa = pack(x), pack(y_sub), pack(y_flat)
        with torch.cuda.device(self.device):
            _check(self.lib.pyrhe_engine_loo_dots_multi(self._h, C.cast(xa, _vp), len(x), C.cast(sa, _vp), len(y_sub),
                                                        C.cast(fa, _vp) if y_flat else None, len(y_flat), 1 if want_self else 0,
                                                        int(period), int(n_sub), int(n_batch), _ptr(out_sub),
                                                        None if out_flat is None else _ptr(out_flat),
                                                        None if out_self is None else _ptr(out_self), _stream_ptr(self.device)),
                   "pyrhe_engine_loo_dots_multi")
        return out_sub, out_flat, out_self

    def profile(self, enable: bool):
        _check(self.lib.pyrhe_engine_profile(self._h, 1 if enable else 0), "pyrhe_engine_profile")

    def profile_read(self):
        a, b, c, n = C.c_double(), C.c_double(), C.c_double(), C.c_int64()
        _check(self.lib.pyrhe_engine_profile_read(self._h, C.byref(a), C.byref(b), C.byref(c), C.byref(n)), "pyrhe_engine_profile_read")
        return {"gemm1_ms": a.value, "mid_ms": b.value, "gemm2_ms": c.value, "launches": n.value}
