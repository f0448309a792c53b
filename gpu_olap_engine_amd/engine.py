"""Device-level host API: a thin, typed wrapper over the C ABI for torch device tensors.

Mirrors the reference's kernel-wrapper crate (gpu-kernel-compiler/src): JoinKernel,
AggregateKernel, FilterKernel plus the memory manager (SlabAllocator, TransferQueue).
torch is only used for device memory and streams -- all compute is in libgolap_hip.so.
"""
import ctypes as C

import torch

from . import _native as N
from ._native import GolapError  # noqa: F401


def _ptr(t):
    return 0 if t is None else t.data_ptr()


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _chk_tensor(t, dtype, name, n=None):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise TypeError(f"{name} must be a CUDA(HIP) tensor")
    if t.dtype != dtype:
        raise TypeError(f"{name} must be {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous")
    if n is not None and t.numel() < n:
        raise ValueError(f"{name} has {t.numel()} elements, need {n}")


def choose_strategy(n_build):
    """JoinKernel::choose_strategy, join_kernel.rs:71-77."""
    return N.load().golap_choose_strategy(n_build)


def compute_launch_params(n_rows):
    """FilterKernel::compute_launch_params, filter_kernel.rs:75-79."""
    g, b = C.c_uint32(0), C.c_uint32(0)
    N.load().golap_compute_launch_params(n_rows, C.byref(g), C.byref(b))
    return g.value, b.value


def agg_identity_value(func):
    """AggFunc::identity_value, aggregate_kernel.rs:27-33."""
    return N.load().golap_agg_identity_value(func)


def parse_cmp_op(op):
    """executor.rs:457-465 operator spellings."""
    return N.load().golap_parse_cmp_op(op.encode())


def default_config():
    cfg = N.EngineConfig()
    N.load().golap_engine_config_default(C.byref(cfg))
    return cfg


class Context:
    """One golap_ctx on the current torch device (one process per GPU)."""

    def __init__(self, config=None, profiling=False, max_gpu_memory=None, provision_sigmas=None, max_slabs_per_class=None):
        self._lib = N.load()
        if not torch.cuda.is_available():
            raise GolapError(N.ERR_NO_DEVICE, "no HIP device visible to torch; this package has no CPU fallback")
        cfg = config if config is not None else default_config()
        cfg.device_id = torch.cuda.current_device()
        if profiling:
            cfg.enable_profiling = int(profiling)      # True / 1: every launch is timed; 2: stand-by launches only while a recent join needed them
        if max_gpu_memory is not None:
            cfg.max_gpu_memory = max_gpu_memory
        if provision_sigmas is not None:
            cfg.provision_sigmas = provision_sigmas
        if max_slabs_per_class is not None:
            cfg.max_slabs_per_class = max_slabs_per_class
        self.config = cfg
        h = C.c_void_p()
        st = self._lib.golap_ctx_create(C.byref(cfg), C.byref(h))
        N.check(None, st)
        self._h = h
        self.device = torch.device("cuda", cfg.device_id)

    def close(self):
        if getattr(self, "_h", None):
            for ref in getattr(self, "_comms", []):      # communicators go first: golap_comm_destroy uses the context's device
                c = ref()
                if c is not None:
                    c.close()
            self._lib.golap_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _ck(self, status):
        N.check(self._h, status)

    # ---- developer / test knobs ------------------------------------------------
    def debug_set(self, name, value):
        """golap_debug_set: experiment variants / forced device paths (value None removes the knob)."""
        self._ck(self._lib.golap_debug_set(self._h, name.encode(), None if value is None else str(value).encode()))

    def knobs(self, **kv):
        """context manager: set knobs for the duration of a block"""
        ctx = self

        class _K:
            def __enter__(self_inner):
                for k, v in kv.items():
                    ctx.debug_set(k, v)

            def __exit__(self_inner, *a):
                for k in kv:
                    ctx.debug_set(k, None)
        return _K()

    # ---- info / stats ------------------------------------------------------
    def device_info(self):
        d = N.DeviceInfo()
        self._ck(self._lib.golap_device_info_get(self._h, C.byref(d)))
        return {f: (getattr(d, f).decode() if isinstance(getattr(d, f), bytes) else getattr(d, f)) for f, _ in d._fields_}

    def stats(self):
        s = N.Stats()
        self._ck(self._lib.golap_stats_get(self._h, C.byref(s)))
        out = {}
        for i in range(s.n):
            k = s.k[i]
            out[k.name.decode()] = {"launches": k.launches, "total_ms": k.total_ms, "bytes": k.bytes}
        return out

    def stats_reset(self):
        self._ck(self._lib.golap_stats_reset(self._h))

    def join_plan(self, n_build, n_owners=1):
        p = N.JoinPlan()
        self._ck(self._lib.golap_join_plan_get(self._h, n_build, n_owners, C.byref(p)))
        return {f: getattr(p, f) for f, _ in p._fields_ if not f.startswith("reserved")}

    def join_scratch_bytes(self, n_build, n_probe):
        return self._lib.golap_join_scratch_bytes(self._h, n_build, n_probe)

    # ---- SlabAllocator / TransferQueue ------------------------------------------
    def slab_alloc(self, size):
        p, c = C.c_uint64(0), C.c_uint32(0)
        self._ck(self._lib.golap_slab_alloc(self._h, size, C.byref(p), C.byref(c)))
        return p.value, c.value

    def slab_free(self, ptr, class_idx):
        self._ck(self._lib.golap_slab_free(self._h, ptr, class_idx))

    def slab_stats(self):
        s = N.SlabStats()
        self._ck(self._lib.golap_slab_get_stats(self._h, C.byref(s)))
        return {"class_size": list(s.class_size)[:s.n_classes], "allocated": list(s.allocated)[:s.n_classes],
                "pooled": list(s.pooled)[:s.n_classes], "bytes_reserved": s.bytes_reserved}

    def xfer_enqueue(self, direction, src, dst, size, stream_id=0):
        self._ck(self._lib.golap_xfer_enqueue(self._h, direction, src, dst, size, stream_id))

    def xfer_synchronize(self):
        self._ck(self._lib.golap_xfer_synchronize(self._h))

    # ---- GpuFilter ------------------------------------------------------------
    def filter_mask(self, column, op, literal, valid=None):
        """FilterKernel::launch -> u8 mask (1 = pass).  valid: uint8 per row, 0 = NULL (never passes)."""
        _chk_tensor(column, torch.int64, "column")
        n = column.numel()
        mask = torch.empty(n, dtype=torch.uint8, device=column.device)
        pred = N.FilterPredicate(0, op if isinstance(op, int) else parse_cmp_op(op), int(literal))
        g, b = compute_launch_params(n)
        prm = N.FilterParams(_ptr(column), n, _ptr(mask), g, b, _ptr(valid))
        self._ck(self._lib.golap_filter_launch(self._h, C.byref(pred), C.byref(prm), _stream()))
        return mask

    def filter_compact(self, column, op, literal, valid=None):
        """-> (indices u32 as int32 tensor view [n], count tensor u64 as int64[1]); indices[:count] are valid."""
        _chk_tensor(column, torch.int64, "column")
        n = column.numel()
        idx = torch.empty(max(n, 1), dtype=torch.int32, device=column.device)
        cnt = torch.zeros(1, dtype=torch.int64, device=column.device)
        pred = N.FilterPredicate(0, op if isinstance(op, int) else parse_cmp_op(op), int(literal))
        self._ck(self._lib.golap_filter_compact_nullable(self._h, C.byref(pred), _ptr(column), _ptr(valid), n, _ptr(idx), _ptr(cnt), _stream()))
        return idx, cnt

    def filter_compact_and(self, columns, ops, literals, valids=None):
        """golap_filter_compact_and: rows for which EVERY comparison columns[k] <ops[k]> literals[k] is true (NULL operands: not true).
        -> (indices, count) as filter_compact."""
        k = len(columns)
        for c in columns:
            _chk_tensor(c, torch.int64, "column")
        n = columns[0].numel()
        idx = torch.empty(max(n, 1), dtype=torch.int32, device=columns[0].device)
        cnt = torch.zeros(1, dtype=torch.int64, device=columns[0].device)
        preds = (N.FilterPredicate * k)(*[N.FilterPredicate(i, o if isinstance(o, int) else parse_cmp_op(o), int(l)) for i, (o, l) in enumerate(zip(ops, literals))])
        cptrs = (C.c_uint64 * k)(*[_ptr(c) for c in columns])
        vptrs = (C.c_uint64 * k)(*[_ptr(v) for v in valids]) if valids is not None else None
        self._ck(self._lib.golap_filter_compact_and(self._h, preds, cptrs, vptrs, k, n, _ptr(idx), _ptr(cnt), _stream()))
        return idx, cnt

    def take(self, column, indices, count=None, n_max=None):
        if column.element_size() != 8:
            raise TypeError("take: 8-byte columns only (executor.rs:496-507)")
        n_max = indices.numel() if n_max is None else n_max
        out = torch.empty(n_max, dtype=column.dtype, device=column.device)
        self._ck(self._lib.golap_take_8byte(self._h, _ptr(column), _ptr(indices), _ptr(count), n_max, _ptr(out), _stream()))
        return out

    def filter_sum(self, key_column, value_column, op, literal, key_valid=None, value_valid=None):
        """-> f64[2] device tensor: (sum of kept values as f64, kept rows)."""
        _chk_tensor(key_column, torch.int64, "key_column")
        n = key_column.numel()
        if value_column.dtype not in (torch.int64, torch.float64) or value_column.numel() < n:
            raise TypeError("value_column must be int64/float64 with >= n rows")
        out = torch.zeros(2, dtype=torch.float64, device=key_column.device)
        pred = N.FilterPredicate(0, op if isinstance(op, int) else parse_cmp_op(op), int(literal))
        self._ck(self._lib.golap_filter_sum_nullable(self._h, C.byref(pred), _ptr(key_column), _ptr(key_valid), _ptr(value_column), _ptr(value_valid),
                                                     1 if value_column.dtype == torch.float64 else 0, n, _ptr(out), _stream()))
        return out

    # ---- GpuHashJoin ------------------------------------------------------------
    def join(self, build_keys, probe_keys, out_pairs=None, max_pairs=None, strategy=None, build_rid_base=0, probe_rid_base=0,
             out_count=None, join_type=0, build_valid=None, probe_valid=None):
        """JoinKernel::join.  -> (pairs int32[max_pairs,2] holding u32 (build_rid, probe_rid), count int64[1] device).

        Pair order is unspecified; the count keeps counting past max_pairs (join_kernels.cuh:155-160).
        Composite keys: build_keys / probe_keys may be lists of up to 4 int64 tensors (rows match when every column is equal).
        NULLs: build_valid / probe_valid = list (one entry per key column) of None or uint8 tensors, 0 = NULL (matches nothing).
        """
        bcols = list(build_keys) if isinstance(build_keys, (list, tuple)) else [build_keys]
        pcols = list(probe_keys) if isinstance(probe_keys, (list, tuple)) else [probe_keys]
        if len(bcols) != len(pcols) or not 1 <= len(bcols) <= 4:
            raise ValueError("1..4 key columns per side, the same number on both sides")
        for t in bcols + pcols:
            _chk_tensor(t, torch.int64, "key column")
        build_keys, probe_keys = bcols[0], pcols[0]
        bval = list(build_valid) if build_valid is not None else [None] * len(bcols)
        pval = list(probe_valid) if probe_valid is not None else [None] * len(pcols)
        nb, npr = build_keys.numel(), probe_keys.numel()
        if strategy is None:
            strategy = choose_strategy(nb)
        if out_pairs is None:
            max_pairs = npr if max_pairs is None else max_pairs
            out_pairs = torch.empty((max(max_pairs, 1), 2), dtype=torch.int32, device=build_keys.device)
        else:
            _chk_tensor(out_pairs, torch.int32, "out_pairs")
            max_pairs = out_pairs.numel() // 2 if max_pairs is None else max_pairs
        if out_count is None:
            out_count = torch.zeros(1, dtype=torch.int64, device=build_keys.device)
        prm = N.JoinParams(_ptr(build_keys), _ptr(probe_keys), nb, npr, _ptr(out_pairs), _ptr(out_count), max_pairs,
                           build_rid_base, probe_rid_base, join_type, len(bcols),
                           (C.c_uint64 * 3)(*([_ptr(t) for t in bcols[1:]] + [0] * (4 - len(bcols)))),
                           (C.c_uint64 * 3)(*([_ptr(t) for t in pcols[1:]] + [0] * (4 - len(pcols)))),
                           (C.c_uint64 * 4)(*([_ptr(t) for t in bval] + [0] * (4 - len(bval)))),
                           (C.c_uint64 * 4)(*([_ptr(t) for t in pval] + [0] * (4 - len(pval)))))
        self._ck(self._lib.golap_join(self._h, strategy, C.byref(prm), _stream()))
        return out_pairs, out_count

    def radix_partition(self, keys, row_ids=None, rid_base=0, n_owners=1, out_keys=None, out_rids=None, extra_bits=0, narrow=False):
        """radix_partition_kernel as a stand-alone step -> (keys, row_ids(int32 view of u32), offsets int32[(256<<extra_bits)+1] by slot).

        narrow=True -> (tuples, None, offsets, wide_flag): the rows as ONE int64 tensor of (key32 | row id << 32) tuples; only
        valid when wide_flag (device int32[1]) reads 0 afterwards (every key in [0, 2^32))."""
        _chk_tensor(keys, torch.int64, "keys")
        n = keys.numel()
        if out_keys is None:
            out_keys = torch.empty(max(n, 1), dtype=torch.int64, device=keys.device)
        if out_rids is None and not narrow:
            out_rids = torch.empty(max(n, 1), dtype=torch.int32, device=keys.device)
        offs = torch.zeros((256 << extra_bits) + 1, dtype=torch.int32, device=keys.device)
        flag = torch.zeros(1, dtype=torch.int32, device=keys.device) if narrow else None
        prm = N.PartitionParams(_ptr(keys), _ptr(row_ids), n, rid_base, n_owners, extra_bits, _ptr(out_keys), _ptr(out_rids), _ptr(offs),
                                1 if narrow else 0, 0, _ptr(flag))
        self._ck(self._lib.golap_radix_partition(self._h, C.byref(prm), _stream()))
        if narrow:
            return out_keys, None, offs, flag
        return out_keys, out_rids, offs

    @staticmethod
    def _side(keys, rids, seg_off, seg_len, rows=None):
        nseg = len(seg_off)
        arr = C.c_uint64 * nseg
        so, sl = arr(*seg_off), arr(*seg_len)
        side = N.PartitionedSide(_ptr(keys), _ptr(rids), keys.numel() if rows is None else rows, so, sl)
        side._keep = (so, sl)   # keep the host tables alive for the duration of the call
        return side

    def join_partitioned(self, b_keys, b_rids, b_seg_off, b_seg_len, p_keys, p_rids, p_seg_off, p_seg_len, n_slots, n_sources,
                         out_pairs, total_build_rows_hint=0, b_rows=None, p_rows=None, out_count=None, pre_bits=8):
        """Join of radix-partitioned (post-exchange) inputs; seg_off/seg_len: host sequences [n_slots*n_sources]."""
        if out_count is None:
            out_count = torch.zeros(1, dtype=torch.int64, device=b_keys.device)
        b = self._side(b_keys, b_rids, b_seg_off, b_seg_len, b_rows)
        p = self._side(p_keys, p_rids, p_seg_off, p_seg_len, p_rows)
        prm = N.JoinPartitionedParams(b, p, n_slots, n_sources, pre_bits, total_build_rows_hint, _ptr(out_pairs), _ptr(out_count),
                                      out_pairs.numel() // 2)
        self._ck(self._lib.golap_join_partitioned(self._h, C.byref(prm), _stream()))
        return out_pairs, out_count

    def join_partitioned_begin(self, n_groups, n_sources, pre_bits, total_build_rows_hint, build_rows, probe_rows, narrow=False):
        pl = N.JoinPartitionedPlan(n_groups, n_sources, pre_bits, total_build_rows_hint, build_rows, probe_rows, 1 if narrow else 0, 0)
        self._ck(self._lib.golap_join_partitioned_begin(self._h, C.byref(pl), _stream()))

    def join_partitioned_narrow_ok(self, n_groups, n_sources, pre_bits, total_build_rows_hint):
        pl = N.JoinPartitionedPlan(n_groups, n_sources, pre_bits, total_build_rows_hint, 0, 0, 1, 0)
        return bool(self._lib.golap_join_partitioned_narrow_ok(C.byref(pl)))

    def join_partitioned_side(self, which, keys, rids, seg_off, seg_len, rows=None):
        side = self._side(keys, rids, seg_off, seg_len, rows)
        self._ck(self._lib.golap_join_partitioned_side(self._h, which, C.byref(side), _stream()))

    def join_partitioned_abort(self):
        self._ck(self._lib.golap_join_partitioned_abort(self._h))

    def join_partitioned_finish(self, out_pairs, out_count=None):
        if out_count is None:
            out_count = torch.zeros(1, dtype=torch.int64, device=out_pairs.device)
        self._ck(self._lib.golap_join_partitioned_finish(self._h, _ptr(out_pairs), _ptr(out_count), out_pairs.numel() // 2, _stream()))
        return out_pairs, out_count

    # ---- multi-GPU sharded join (golap_comm / golap_sharded_join) -----------------------
    def comm_create_rccl(self, comm_id, world, rank):
        """Built-in RCCL communicator; comm_id = the 128 bytes of comm_unique_id() made on one rank and handed to all."""
        buf = (C.c_uint8 * N.COMM_ID_BYTES).from_buffer_copy(bytes(comm_id))
        h = C.c_void_p()
        self._ck(self._lib.golap_comm_create_rccl(self._h, buf, world, rank, C.byref(h)))
        return Comm(self, h, world, rank)

    def comm_create_custom(self, all_gather, exchange, world, rank):
        """Communicator over host-supplied transport callbacks (N.ALL_GATHER_FN / N.EXCHANGE_FN signatures)."""
        tr = N.Transport(None, N.ALL_GATHER_FN(all_gather), N.EXCHANGE_FN(exchange))
        h = C.c_void_p()
        self._ck(self._lib.golap_comm_create_custom(self._h, C.byref(tr), world, rank, C.byref(h)))
        c = Comm(self, h, world, rank)
        c._keep = tr      # the C side copied the struct; the callback objects must stay alive
        return c

    # ---- GpuAggregate -----------------------------------------------------------
    def aggregate(self, specs, values, group_keys=None, n_groups=1, want_counts=False, hashed=False, value_valid=None, _comm=None):
        """AggregateKernel::aggregate.

        specs  : list of (func, value_col_index) -- output plane i belongs to specs[i]
        values : list of int64/float64 device tensors
        -> f64[n_specs, n_groups] (and u64 counts as int64[n_groups] when want_counts)
        hashed : GOLAP_KEYS_HASHED -- any keys; n_groups bounds the number of distinct keys;
                 -> (out, [counts,] slot_keys int64[n_groups], n_distinct int32[1] device)
        value_valid : None or a list (one entry per value column) of None | uint8 device tensors, 0 = NULL: SQL aggregates skip them
        """
        dev = values[0].device
        n_rows = values[0].numel()
        nv = len(values)
        vptrs = (C.c_uint64 * nv)(*[_ptr(v) for v in values])
        vdt = (C.c_int32 * nv)(*[N.DT_FLOAT64 if v.dtype == torch.float64 else N.DT_INT64 for v in values])
        for v in values:
            if v.dtype not in (torch.int64, torch.float64):
                raise TypeError("aggregate inputs must be int64 or float64 (executor.rs:520-526)")
            if v.numel() != n_rows:
                raise ValueError("value columns differ in length")
        sp = (N.AggregateSpec * len(specs))(*[N.AggregateSpec(f, c, i) for i, (f, c) in enumerate(specs)])
        out = torch.zeros((len(specs), n_groups), dtype=torch.float64, device=dev)
        counts = torch.zeros(n_groups, dtype=torch.int64, device=dev) if want_counts else None
        kd = N.DT_INT64
        if group_keys is not None:
            if group_keys.dtype == torch.int32:
                kd = N.DT_INT32
            elif group_keys.dtype != torch.int64:
                raise TypeError("group keys must be int32 or int64")
            if group_keys.numel() != n_rows:
                raise ValueError("group key column differs in length")
        okeys = torch.zeros(n_groups, dtype=torch.int64, device=dev) if hashed else None
        ong = torch.zeros(1, dtype=torch.int32, device=dev) if hashed else None
        vvp = None
        if value_valid is not None:
            if len(value_valid) != nv:
                raise ValueError("value_valid needs one entry per value column")
            for v in value_valid:
                if v is not None and (v.dtype != torch.uint8 or v.numel() != n_rows):
                    raise TypeError("validity columns are uint8 with one byte per row")
            vvp = (C.c_uint64 * nv)(*[_ptr(v) for v in value_valid])
        prm = N.AggregateParams(_ptr(group_keys), vptrs, nv, n_rows, n_groups, _ptr(out), kd, vdt, _ptr(counts),
                                N.KEYS_HASHED if hashed else N.KEYS_DENSE, 0, _ptr(okeys), _ptr(ong), vvp)
        if _comm is not None:   # collective form over row-sharded inputs (Comm.sharded_aggregate)
            self._ck(self._lib.golap_sharded_aggregate(self._h, _comm._h, sp, len(specs), C.byref(prm), _stream()))
        else:
            self._ck(self._lib.golap_aggregate(self._h, sp, len(specs), C.byref(prm), _stream()))
        if hashed:
            return (out, counts, okeys, ong) if want_counts else (out, okeys, ong)
        return (out, counts) if want_counts else out

    # ---- arrow-interop widening ----------------------------------------------------
    _WIDEN = {torch.int8: N.WIDEN_I8, torch.int16: N.WIDEN_I16, torch.int32: N.WIDEN_I32, torch.int64: N.WIDEN_I64,
              torch.uint8: N.WIDEN_U8, torch.float32: N.WIDEN_F32, torch.float64: N.WIDEN_F64}

    def widen(self, column, validity=None, in_dtype=None):
        code = self._WIDEN[column.dtype] if in_dtype is None else in_dtype
        out_dt = torch.float64 if code in (N.WIDEN_F32, N.WIDEN_F64) else torch.int64
        out = torch.empty(column.numel(), dtype=out_dt, device=column.device)
        self._ck(self._lib.golap_widen_to_8byte(self._h, _ptr(column), code, _ptr(validity), column.numel(), _ptr(out), _stream()))
        return out

    def hash_utf8(self, offsets, data, valid=None):
        """golap_hash_utf8: FNV-1a 64 of every value of a Utf8 column (offsets int64[n + 1], data uint8 bytes; NULL rows -> 0) -> int64[n]."""
        _chk_tensor(offsets, torch.int64, "offsets")
        n = offsets.numel() - 1
        out = torch.empty(max(n, 0), dtype=torch.int64, device=offsets.device)
        self._ck(self._lib.golap_hash_utf8(self._h, _ptr(offsets), _ptr(data), _ptr(valid), max(n, 0), _ptr(out), _stream()))
        return out

    # ---- device-side ingest / materialisation (ABI 4) ---------------------------------
    def expand_validity(self, bitmap, n_rows, bit_offset=0):
        """golap_expand_validity: Arrow validity bitmap (uint8 tensor, LSB first, first row at bit `bit_offset`) -> uint8[n_rows], 1 = valid"""
        _chk_tensor(bitmap, torch.uint8, "bitmap")
        out = torch.empty(n_rows, dtype=torch.uint8, device=bitmap.device)
        self._ck(self._lib.golap_expand_validity(self._h, _ptr(bitmap), bit_offset, n_rows, _ptr(out), _stream()))
        return out

    def pack_validity(self, valid):
        """golap_pack_validity: uint8[n] (0 = NULL) -> (Arrow bitmap uint8[ceil(n / 8)], NULL count as int64[1] on the device)"""
        _chk_tensor(valid, torch.uint8, "valid")
        n = valid.numel()
        out = torch.empty((n + 7) // 8, dtype=torch.uint8, device=valid.device)
        nulls = torch.empty(1, dtype=torch.int64, device=valid.device)
        self._ck(self._lib.golap_pack_validity(self._h, _ptr(valid), n, _ptr(out), _ptr(nulls), _stream()))
        return out, nulls

    def take_validity(self, src_valid, indices, count=None, n_max=None):
        """golap_take_1byte: validity of gathered rows (src_valid None = the source has no NULLs; index 0xFFFFFFFF -> 0)"""
        n_max = indices.numel() if n_max is None else n_max
        out = torch.zeros(n_max, dtype=torch.uint8, device=indices.device)
        self._ck(self._lib.golap_take_1byte(self._h, _ptr(src_valid), _ptr(indices), _ptr(count), n_max, _ptr(out), _stream()))
        return out

    def pairs_unzip(self, pairs, n_pairs=None):
        """golap_pairs_unzip: int32[m, 2] (build_rid, probe_rid) -> (build ids int32[m], probe ids int32[m])"""
        m = pairs.shape[0] if n_pairs is None else n_pairs
        b = torch.empty(m, dtype=torch.int32, device=pairs.device)
        p = torch.empty(m, dtype=torch.int32, device=pairs.device)
        self._ck(self._lib.golap_pairs_unzip(self._h, _ptr(pairs), m, _ptr(b), _ptr(p), _stream()))
        return b, p

    def sort_u64(self, words, n=None):
        """golap_sort_u64: ascending in-place sort of the first n 64-bit words of `words` (compared as UNSIGNED; int64 or an int32[m, 2] pair list)"""
        n = (words.numel() * words.element_size()) // 8 if n is None else n
        self._ck(self._lib.golap_sort_u64(self._h, _ptr(words), n, _stream()))
        return words

    def key_minmax(self, column, valid=None):
        """golap_key_minmax -> int64[3] on the device: min, max, number of the valid rows"""
        _chk_tensor(column, torch.int64, "column")
        out = torch.empty(3, dtype=torch.int64, device=column.device)
        self._ck(self._lib.golap_key_minmax(self._h, _ptr(column), _ptr(valid), column.numel(), _ptr(out), _stream()))
        return out

    def pack_group_keys(self, columns, valids, mins, weights, null_digits):
        """golap_pack_group_keys: out[i] = sum_c (valid ? col_c[i] - mins[c] : null_digits[c]) * weights[c]"""
        p = N.PackKeysParams()
        p.n_cols, p.n_rows = len(columns), columns[0].numel()
        out = torch.empty(p.n_rows, dtype=torch.int64, device=columns[0].device)
        for c, col in enumerate(columns):
            _chk_tensor(col, torch.int64, "key column")
            p.col_ptrs[c], p.validity_ptrs[c] = _ptr(col), _ptr(valids[c]) if valids else 0
            p.min_value[c], p.weight[c], p.null_digit[c] = int(mins[c]), int(weights[c]), int(null_digits[c])
        p.out_ptr = _ptr(out)
        self._ck(self._lib.golap_pack_group_keys(self._h, C.byref(p), _stream()))
        return out

    def device_copy(self, src, dst, nbytes):
        self._ck(self._lib.golap_device_copy(self._h, _ptr(src), _ptr(dst), nbytes, _stream()))

    def device_fill_u8(self, dst, value, nbytes):
        self._ck(self._lib.golap_device_fill_u8(self._h, _ptr(dst), value, nbytes, _stream()))

    # ---- generators / verifier ---------------------------------------------------
    def generate(self, kind, count, modulus=1, seed=0, first=0, out=None):
        dt = {N.GEN_BUILD_KEYS: torch.int64, N.GEN_SM64_MOD_I64: torch.int64, N.GEN_SM64_MOD_I32: torch.int32,
              N.GEN_SM64_MOD_F64: torch.float64, N.GEN_SHIFT32_I64: torch.int64}[kind]
        if out is None:
            out = torch.empty(count, dtype=dt, device=self.device)
        self._ck(self._lib.golap_generate(self._h, kind, seed, first, count, modulus, _ptr(out), _stream()))
        return out

    def verify_pairs(self, pairs, n_pairs, build_keys, probe_keys, build_rid_base=0, probe_rid_base=0):
        """-> int64[4] device: equal-key pairs, checksum, sum of probe ids, xor of mixed probe ids (u64 bit patterns)."""
        out = torch.zeros(4, dtype=torch.int64, device=pairs.device)
        self._ck(self._lib.golap_verify_pairs(self._h, _ptr(pairs), n_pairs, _ptr(build_keys), _ptr(probe_keys),
                                              build_rid_base, probe_rid_base, _ptr(out), _stream()))
        return out

    def synchronize(self):
        self._ck(self._lib.golap_ctx_synchronize(self._h, _stream()))


def comm_unique_id():
    """ncclGetUniqueId through the library (rank 0 calls it, the host hands the 128 bytes to every rank)."""
    buf = (C.c_uint8 * N.COMM_ID_BYTES)()
    N.check(None, N.load().golap_comm_unique_id(buf))
    return bytes(buf)


def shard_range(n_total, rank, world):
    """golap_shard_range: rank g holds rows [g*N/P, (g+1)*N/P) (last rank takes the remainder)."""
    lo, hi = C.c_uint64(0), C.c_uint64(0)
    N.load().golap_shard_range(n_total, rank, world, C.byref(lo), C.byref(hi))
    return lo.value, hi.value


def sharded_plan(world, n_build_total, n_probe_total=None, max_msg_bytes=0):
    """golap_sharded_plan_get (host only): the fast path's provisioned bucket ranges and message sizes."""
    pl = N.ShardedPlan()
    N.check(None, N.load().golap_sharded_plan_get(world, n_build_total, n_build_total if n_probe_total is None else n_probe_total,
                                                  None, None, max_msg_bytes, C.byref(pl)))
    return {"eligible": bool(pl.eligible), "extra_bits": pl.extra_bits, "n_buckets": pl.n_buckets, "buckets_per_rank": pl.buckets_per_rank,
            "local_bits": pl.local_bits, "sub_ranges": pl.sub_ranges, "chunks": pl.chunks,
            "cap_rows": [[pl.cap_rows[s][r] for r in range(world)] for s in range(2)],
            "cap_rows_last": [[pl.cap_rows_last[s][r] for r in range(world)] for s in range(2)],
            "msg_bytes": [[pl.msg_bytes[s][r] for r in range(world)] for s in range(2)],
            "msg_pieces": [[pl.msg_pieces[s][r] for r in range(world)] for s in range(2)]}


class Comm:
    """golap_comm: a context bound to a communicator (one process per GPU)."""

    def __init__(self, ctx, handle, world, rank):
        import weakref
        self.ctx, self._h, self.world, self.rank = ctx, handle, world, rank
        if not hasattr(ctx, "_comms"):
            ctx._comms = []
        ctx._comms.append(weakref.ref(self))

    def close(self):
        if getattr(self, "_h", None):
            self.ctx._lib.golap_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sharded_join(self, build_keys, probe_keys, build_rid_base, probe_rid_base, n_build_total, n_probe_total=None, out_pairs=None,
                     max_pairs=None, out_count=None, join_type=0, force_exact=False, max_msg_bytes=0, build_rows_per_rank=None,
                     probe_rows_per_rank=None, build_valid=None, probe_valid=None):
        """golap_sharded_join (collective).  -> (pairs int32[max_pairs,2] with GLOBAL u32 row ids, count int64[1] device).
        *_rows_per_rank: the row count of every rank's shard (sequences of world ints) when the split is not shard_range's even one --
        every rank must pass the same lists (the fast path sizes its messages from them without asking anybody).
        build_keys / probe_keys: one int64 tensor, or a list of up to 4 (a composite key); *_valid: None or a list (one entry per key
        column) of None | uint8 tensors, 0 = NULL (such rows match nothing)."""
        bcols = list(build_keys) if isinstance(build_keys, (list, tuple)) else [build_keys]
        pcols = list(probe_keys) if isinstance(probe_keys, (list, tuple)) else [probe_keys]
        if len(bcols) != len(pcols) or not 1 <= len(bcols) <= 4:
            raise ValueError("1 to 4 key columns, the same number on both sides")
        for t in bcols + pcols:
            _chk_tensor(t, torch.int64, "key column")
        build_keys, probe_keys = bcols[0], pcols[0]
        if n_probe_total is None:
            n_probe_total = n_build_total
        if out_pairs is None:
            max_pairs = max(1, probe_keys.numel() * 2 if max_pairs is None else max_pairs)
            out_pairs = torch.empty((max_pairs, 2), dtype=torch.int32, device=build_keys.device)
        elif max_pairs is None:
            max_pairs = out_pairs.numel() // 2
        if out_count is None:
            out_count = torch.zeros(1, dtype=torch.int64, device=build_keys.device)
        def rows(v, mine):
            if v is None:
                return None
            if len(v) != self.world or int(v[self.rank]) != mine:
                raise ValueError("rows_per_rank needs one entry per rank, and this rank's entry must be its shard's row count")
            return (C.c_uint64 * self.world)(*[int(x) for x in v])
        brows, prows = rows(build_rows_per_rank, build_keys.numel()), rows(probe_rows_per_rank, probe_keys.numel())
        prm = N.ShardedJoinParams(_ptr(build_keys), _ptr(probe_keys), build_keys.numel(), probe_keys.numel(), build_rid_base, probe_rid_base,
                                  n_build_total, n_probe_total, brows, prows, _ptr(out_pairs), _ptr(out_count), max_pairs, join_type,
                                  1 if force_exact else 0, max_msg_bytes)
        prm.n_key_cols = len(bcols)
        for c in range(1, len(bcols)):
            prm.build_key_extra[c - 1] = _ptr(bcols[c])
            prm.probe_key_extra[c - 1] = _ptr(pcols[c])
        for side_valid, dst in ((build_valid, prm.build_validity), (probe_valid, prm.probe_validity)):
            if side_valid is not None:
                for c, v in enumerate(side_valid):
                    if v is not None:
                        _chk_tensor(v, torch.uint8, "validity column")
                        dst[c] = _ptr(v)
        self.ctx._ck(self.ctx._lib.golap_sharded_join(self.ctx._h, self._h, C.byref(prm), _stream()))
        return out_pairs, out_count

    def sharded_aggregate(self, specs, values, group_keys=None, n_groups=1, want_counts=False, hashed=False, value_valid=None):
        """golap_sharded_aggregate (collective): Context.aggregate over THIS rank's rows of a row-sharded table.  Dense keys: the result
        is replicated on every rank; hashed keys: it stays sharded (every key is reported by exactly one rank)."""
        return self.ctx.aggregate(specs, values, group_keys=group_keys, n_groups=n_groups, want_counts=want_counts, hashed=hashed,
                                  value_valid=value_valid, _comm=self)

    def selftest(self, nbytes):
        """golap_comm_selftest -> (wrong 4-byte words, index of the first one or None)."""
        bad, first = C.c_uint64(0), C.c_uint64(0)
        self.ctx._ck(self.ctx._lib.golap_comm_selftest(self.ctx._h, self._h, nbytes, C.byref(bad), C.byref(first)))
        return bad.value, (first.value if bad.value else None)

    def info(self):
        i = N.ShardedJoinInfo()
        self.ctx._ck(self.ctx._lib.golap_sharded_join_info_get(self._h, C.byref(i)))
        return {"fast_path": bool(i.fast_path), "fast_path_abandoned": bool(i.fast_path_abandoned), "extra_bits": i.extra_bits,
                "n_groups": i.n_groups, "local_bits": i.local_bits, "host_syncs": i.host_syncs,
                "rows_received": list(i.rows_received), "bytes_sent": list(i.bytes_sent), "chunks": i.chunks, "sub_ranges": i.sub_ranges, "halves": i.halves,
                # device timestamps of the fast path's stages, ms since the call's first kernel: [begin, end] (profiling contexts only)
                "stage_ms": {nm: [round(i.stage_ms[k][0], 4), round(i.stage_ms[k][1], 4)] for k, nm in enumerate(N.SHARD_STAGES)},
                "total_ms": round(i.total_ms, 4)}
