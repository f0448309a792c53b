"""SlabAllocator / TransferQueue on a real device (the reference's own tests need a CUDA device too,
transfer_queue.rs:146-153, and its allocator test is empty, slab_allocator.rs:173-181)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
from gpu_olap_engine_amd import _native as N  # noqa: E402
from gpu_olap_engine_amd.engine import Context, GolapError  # noqa: E402


def test_slab_size_classes_and_reuse():
    with Context(max_gpu_memory=0) as c:
        p1, k1 = c.slab_alloc(100)                 # -> 1 MiB class (first class >= size, slab_allocator.rs:95-107)
        p2, k2 = c.slab_alloc((1 << 20) + 1)       # -> 4 MiB class
        assert k1 == 0 and k2 == 1 and p1 != p2
        st = c.slab_stats()
        assert st["class_size"][:5] == [1 << 20, 4 << 20, 16 << 20, 64 << 20, 256 << 20]   # README.md:142
        assert st["allocated"][0] == 1 and st["allocated"][1] == 1
        c.slab_free(p1, k1)
        assert c.slab_stats()["pooled"][0] == 1
        p3, k3 = c.slab_alloc(4096)                # reuses the pooled slab (slab_allocator.rs:57-60)
        assert (p3, k3) == (p1, k1)
        c.slab_free(p3, k3)
        c.slab_free(p2, k2)
        with pytest.raises(GolapError) as e:       # slab_allocator.rs:72-74
            c.slab_free(p2, 99)
        assert e.value.code == N.ERR_INVALID_ARG
        with pytest.raises(GolapError) as e:       # above the largest class (:102-106)
            c.slab_alloc(1 << 50)
        assert e.value.code == N.ERR_OOM


def test_max_gpu_memory_cap():
    with Context(max_gpu_memory=8 << 20) as c:
        a = c.slab_alloc(1 << 20)
        b = c.slab_alloc(4 << 20)
        with pytest.raises(GolapError) as e:
            c.slab_alloc(16 << 20)
        assert e.value.code == N.ERR_OOM
        c.slab_free(*a)
        c.slab_free(*b)


def test_transfer_queue_roundtrip_pageable_and_pinned():
    with Context(max_gpu_memory=0) as c:
        n = 5_000_017     # > 2 bounce buffers of 8 MiB
        src = np.arange(n, dtype=np.int64)
        dst = np.zeros(n, dtype=np.int64)
        d = torch.empty(n, dtype=torch.int64, device="cuda")
        half = n // 2 * 8
        # two streams, round-robin by stream_id % num_streams (transfer_queue.rs:72)
        c.xfer_enqueue(N.HOST_TO_DEVICE, src.ctypes.data, d.data_ptr(), half, 0)
        c.xfer_enqueue(N.HOST_TO_DEVICE, src.ctypes.data + half, d.data_ptr() + half, n * 8 - half, 9)
        c.xfer_synchronize()
        assert torch.equal(d.cpu(), torch.from_numpy(src))
        c.xfer_enqueue(N.DEVICE_TO_HOST, d.data_ptr(), dst.ctypes.data, n * 8, 3)
        c.xfer_synchronize()
        assert np.array_equal(dst, src)
        pin = torch.empty(n, dtype=torch.int64).pin_memory()
        c.xfer_enqueue(N.DEVICE_TO_HOST, d.data_ptr(), pin.data_ptr(), n * 8, 1)
        c.xfer_synchronize()
        assert np.array_equal(pin.numpy(), src)
        with pytest.raises(GolapError):
            c.xfer_enqueue(7, src.ctypes.data, d.data_ptr(), 8, 0)


def test_device_info_is_mi355x(ctx):
    info = ctx.device_info()
    assert info["wavefront_size"] == 64
    assert "gfx950" in info["gcn_arch"]
    assert info["compute_units"] >= 200


def test_workspace_is_guarded_while_a_partitioned_join_is_open(ctx):
    """golap_join_partitioned_begin keeps pointers into the context's cached workspace across calls: until finish / abort every
    other compute call on that context must fail loudly instead of overwriting the partitioned rows (golap.h)."""
    import numpy as np
    import torch
    from gpu_olap_engine_amd.engine import GolapError
    k = torch.arange(100_000, dtype=torch.int64, device="cuda")
    ctx.join_partitioned_begin(256, 1, 8, 100_000, 100_000, 100_000)
    try:
        with pytest.raises(GolapError) as e:
            ctx.radix_partition(k)
        assert "workspace" in str(e.value)
        with pytest.raises(GolapError):
            ctx.join(k, k)
        with pytest.raises(GolapError):
            ctx.join_partitioned_begin(256, 1, 8, 100_000, 100_000, 100_000)      # a second begin is refused as well
    finally:
        ctx.join_partitioned_abort()
    pairs, cnt = ctx.join(k, k)                                                   # usable again, and the sticky error is gone
    torch.cuda.synchronize()
    assert int(cnt.item()) == 100_000
