"""Generates the committed golden fixtures under tests/golden/.

The reference cannot be built or run here (Rust + CUDA toolchains absent), so the
vectors come from (a) the reference's own unit tests (values copied as DATA with their file:line),
(b) hand evaluation of join_kernels.cuh:26-41 done in the survey session, and
(c) the independent numpy twin (tests/np_twin.py) for join pairs / filters / aggregates -- the C++
oracle is then checked against all three.  Run from the repo root:  python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import np_twin as T  # noqa: E402

# (a)+(b): known-answer vectors
kat = {
    "source": "SURVEY.md 8(c): hand-evaluated from gpu-kernel-compiler/kernels/join_kernels.cuh:26-41, :69",
    "hash_int64": {"0": 0x00000000, "1": 0x514E28B7, "2": 0x30F4C306, "42": 0x087FCD5C, "255": 0x6C63D583, "256": 0x4570315F,
                   "1000000": 0x02409889, "4294967296": 0x514E28B7, "4294967297": 0x00000000, "-1": 0x00000000,
                   "-2": 0x514E28B7, "9223372036854775807": 0x6D3C65A0, "-9223372036854775808": 0x6D3C65A0,
                   "81985529216486895": 0x1D6EF8AE},
    "hash_int32": {"0": 0, "1": 0x514E28B7, "-1": 0x81F16F39, "42": 0x087FCD5C},
    "partition_shift0_shift24": {"1": [183, 81], "42": [92, 8], "1000000": [137, 2]},
    "dense_2pow20_partition_sizes": {"min": 3935, "max": 4279, "mean": 4096},
    "reference_unit_tests": {
        "executor.rs:553-570 test_eval_filter_mask_gt": {"col": [5, 15, 25], "op": ">", "literal": 10, "mask": [0, 1, 1]},
        "join_kernel.rs:84-88 strategy_selection": {"500000": "BroadcastHashJoin", "5000000": "RadixHashJoin"},
        "filter_kernel.rs:86-95 launch_params": {"1024": [4, 256], "1": [1, 256]},
        "aggregate_kernel.rs:138-143 identity_values": {"Sum": 0, "Min": 9223372036854775807, "Max": -9223372036854775808},
        "record_batch_convert.rs:211-217 int32_widens_to_i64": {"int32": [1, 2, 3, 4, 5], "first_i64_le_bytes": [1, 0, 0, 0, 0, 0, 0, 0]},
    },
}
json.dump(kat, open(os.path.join(HERE, "kat.json"), "w"), indent=1)

# (c) join: mixed duplicates / misses / negative / fold-colliding keys, 2 batches per side
rng = np.random.default_rng(20261004)
build = np.concatenate([rng.integers(-500, 500, 3000), np.array([0, -1, 1 << 32, (1 << 32) + 1, -2, 1, 7, 7, 7], np.int64),
                        np.arange(0, 64, dtype=np.int64) << 32]).astype(np.int64)
probe = np.concatenate([rng.integers(-600, 600, 4000), np.array([7, 1 << 32, 5 << 32, -(1 << 63), (1 << 63) - 1], np.int64)]).astype(np.int64)
pairs = T.hash_join(build, probe)
np.savez_compressed(os.path.join(HERE, "join_small.npz"), build=build, probe=probe, pairs=pairs,
                    build_batch_lens=np.array([1000, build.size - 1000], np.uint64), probe_batch_lens=np.array([4000, probe.size - 4000], np.uint64),
                    hash_build=T.hash_int64(build), part_build=T.partition_id(T.hash_int64(build)))

# filter + aggregate
col = rng.integers(-100, 100, 5000).astype(np.int64)
val_i = rng.integers(-1000, 1000, 5000).astype(np.int64)
val_f = (rng.integers(0, 1 << 20, 5000) / 64.0).astype(np.float64)       # dyadic rationals: sums exact in any order
masks = {op: T.filter_mask(col, op, 13) for op in ["=", "==", "!=", "<>", ">", ">=", "<", "<=", "AND"]}
gk = rng.integers(0, 37, 5000).astype(np.int64)
gsum = np.array([val_f[gk == g].sum() for g in range(37)])
gcnt = np.array([(gk == g).sum() for g in range(37)], np.float64)
gmin = np.array([val_f[gk == g].min() for g in range(37)])
gmax = np.array([val_f[gk == g].max() for g in range(37)])
np.savez_compressed(os.path.join(HERE, "filter_agg_small.npz"), col=col, val_i=val_i, val_f=val_f, literal=np.int64(13),
                    **{"mask_" + {"=": "eq1", "==": "eq2", "!=": "ne1", "<>": "ne2", ">": "gt", ">=": "ge", "<": "lt", "<=": "le", "AND": "unknown"}[k]: v
                       for k, v in masks.items()},
                    sum_i_lt=np.float64(val_i[col < 13].astype(np.float64).sum()), kept_lt=np.int64((col < 13).sum()),
                    agg_i=np.array([val_i.astype(np.float64).sum(), val_i.size, val_i.min(), val_i.max(), val_i.astype(np.float64).sum() / val_i.size]),
                    gk=gk, gsum=gsum, gcnt=gcnt, gmin=gmin, gmax=gmax)
# generator vectors (sm64 / build / probe keys) pinned from the numpy twin
i = np.arange(0, 64, dtype=np.uint64)
np.savez_compressed(os.path.join(HERE, "generators.npz"), sm64_seed43=T.sm64(43, i), sm64_seed7_first1000=T.sm64(7, i + np.uint64(1000)),
                    build_keys_N1000003=((np.uint64(2654435761) * i + np.uint64(12345)) % np.uint64(1000003)).astype(np.int64))
print("golden fixtures written")
