"""The device half of the operator layer's plumbing (csrc/golap_ingest.hip, ABI 4) through the C ABI, against numpy / pyarrow:
what record_batch_convert.rs:35-110 and :146-178 do per value on the host -- validity bitmaps to bytes and back, gathers of validity,
pair lists split and ordered, exact key ranges, composite GROUP BY keys."""
import numpy as np
import pyarrow as pa
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def cu(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("n,off", [(1, 0), (7, 0), (8, 3), (63, 5), (64, 0), (1000, 7), (100_003, 1), (1 << 20, 6)])
def test_expand_validity_matches_arrow(ctx, n, off):
    """bit (off + i) of the bitmap, LSB first == pyarrow's is_valid(i) of the sliced array (record_batch_convert.rs:36-40)"""
    rng = np.random.default_rng(n + off)
    vals = [None if x < 0.3 else 1 for x in rng.random(n + off)]
    vals[off + n // 2] = None                                            # (at least one NULL: pyarrow builds no bitmap otherwise)
    arr = pa.array(vals, pa.int64()).slice(off, n)
    bitmap = np.frombuffer(arr.buffers()[0], dtype=np.uint8)
    got = ctx.expand_validity(cu(bitmap.copy()), n, bit_offset=arr.offset).cpu().numpy()
    exp = np.array([v is not None for v in vals[off:]], dtype=np.uint8)
    assert np.array_equal(got, exp)


@pytest.mark.parametrize("n", [1, 8, 9, 64, 1001, 1 << 20, (1 << 20) + 5])
def test_pack_validity_round_trip(ctx, n):
    """bytes -> bitmap: the bitmap pyarrow would build, unused bits of the last byte 0, the NULL count exact; and back"""
    rng = np.random.default_rng(n)
    v = (rng.random(n) > 0.2).astype(np.uint8) * rng.integers(1, 255, n).astype(np.uint8)     # any non-zero byte is "valid"
    bitmap, nulls = ctx.pack_validity(cu(v))
    exp = np.packbits(v != 0, bitorder="little")
    assert np.array_equal(bitmap.cpu().numpy(), exp) and int(nulls.item()) == int((v == 0).sum())
    assert np.array_equal(ctx.expand_validity(bitmap, n).cpu().numpy(), (v != 0).astype(np.uint8))


def test_take_validity(ctx):
    rng = np.random.default_rng(3)
    n, m = 50_000, 120_001
    src = (rng.random(n) > 0.1).astype(np.uint8)
    idx = rng.integers(0, n, m).astype(np.uint32)
    idx[rng.random(m) < 0.05] = 0xFFFFFFFF                              # the missing side of an outer-join row
    got = ctx.take_validity(cu(src), cu(idx.view(np.int32))).cpu().numpy()
    exp = np.where(idx == 0xFFFFFFFF, 0, src[np.minimum(idx, n - 1)]).astype(np.uint8)
    assert np.array_equal(got, exp)
    got = ctx.take_validity(None, cu(idx.view(np.int32))).cpu().numpy()          # a source without NULLs: only the NULL ids are invalid
    assert np.array_equal(got, (idx != 0xFFFFFFFF).astype(np.uint8))
    cnt = torch.tensor([1000], dtype=torch.int64, device="cuda")                  # device-side count: rows beyond it are left alone
    got = ctx.take_validity(cu(src), cu(idx.view(np.int32)), count=cnt).cpu().numpy()
    assert np.array_equal(got[:1000], exp[:1000]) and not got[1000:].any()


@pytest.mark.parametrize("m", [1, 2, 1000, 3_000_001])
def test_sort_and_unzip_pairs(ctx, m):
    """a pair list sorted as 64-bit words is in the canonical (probe_rid, build_rid) order, NULL ids last; unzip splits it"""
    rng = np.random.default_rng(m)
    pairs = rng.integers(0, 1 << 32, (m, 2), dtype=np.uint64).astype(np.uint32)
    pairs[rng.random(m) < 0.01, 0] = 0xFFFFFFFF
    pairs[rng.random(m) < 0.01, 1] = 0xFFFFFFFF
    d = cu(pairs.view(np.int32))
    ctx.sort_u64(d)
    b, p = ctx.pairs_unzip(d)
    order = np.lexsort((pairs[:, 0], pairs[:, 1]))
    assert np.array_equal(d.cpu().numpy().view(np.uint32), pairs[order])
    assert np.array_equal(b.cpu().numpy().view(np.uint32), pairs[order, 0]) and np.array_equal(p.cpu().numpy().view(np.uint32), pairs[order, 1])


def test_key_minmax_is_exact_and_null_aware(ctx):
    rng = np.random.default_rng(9)
    n = 1_000_003
    k = rng.integers(-(1 << 62), 1 << 62, n)
    k[5] = np.iinfo(np.int64).max - 1                     # beyond 2^53: a double would round it
    k[7] = np.iinfo(np.int64).min + 1
    v = (rng.random(n) > 0.1).astype(np.uint8)
    v[5] = v[7] = 1
    assert ctx.key_minmax(cu(k)).cpu().tolist() == [int(k.min()), int(k.max()), n]
    assert ctx.key_minmax(cu(k), cu(v)).cpu().tolist() == [int(k[v == 1].min()), int(k[v == 1].max()), int(v.sum())]
    v[:] = 0
    assert ctx.key_minmax(cu(k), cu(v)).cpu().tolist() == [np.iinfo(np.int64).max, np.iinfo(np.int64).min, 0]


def test_pack_group_keys(ctx):
    rng = np.random.default_rng(4)
    n = 300_000
    a, b, c = rng.integers(-5, 9, n), rng.integers(1000, 1007, n), rng.integers(-(1 << 40), -(1 << 40) + 50, n)
    va, vc = (rng.random(n) > 0.1).astype(np.uint8), (rng.random(n) > 0.5).astype(np.uint8)
    mins = [int(a.min()), int(b.min()), int(c.min())]
    digits = [int(a.max()) - mins[0] + 2, int(b.max()) - mins[1] + 1, int(c.max()) - mins[2] + 2]     # + 1 for the NULL digit where NULLs occur
    weights = [digits[1] * digits[2], digits[2], 1]
    nulld = [digits[0] - 1, 0, digits[2] - 1]
    got = ctx.pack_group_keys([cu(a), cu(b), cu(c)], [cu(va), None, cu(vc)], mins, weights, nulld).cpu().numpy()
    da = np.where(va == 1, a - mins[0], nulld[0])
    dc = np.where(vc == 1, c - mins[2], nulld[2])
    assert np.array_equal(got, da * weights[0] + (b - mins[1]) * weights[1] + dc)


def test_device_copy_and_fill(ctx):
    src = torch.arange(1000, dtype=torch.int64, device="cuda")
    dst = torch.zeros(3000, dtype=torch.int64, device="cuda")
    ctx.device_copy(src, dst[1000:], 8000)
    v = torch.zeros(77, dtype=torch.uint8, device="cuda")
    ctx.device_fill_u8(v[3:], 1, 70)
    torch.cuda.synchronize()
    assert torch.equal(dst[1000:2000], src) and not dst[:1000].any() and not dst[2000:].any()
    assert v.cpu().tolist() == [0] * 3 + [1] * 70 + [0] * 4
