"""The entry point that allocates nothing (cfd_resample_nearest_noalloc, include/cfd_amd.h): same results, bit for
bit, as cfd_resample_nearest with a caller-supplied workspace -- on two calls in a row, the second with a smaller
input (the built-in scratch keeps the first call's contents) -- and sizes beyond the built-in buffer are refused.
Collected last on purpose: the round's end runs `pytest -x`."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_noalloc_equals_workspace_variant_twice():
    from cfd_amd import _lib, ops, synthetic
    big = [synthetic.scattered_case(s, n_z=101, n_y=9 + 2 * s, jitter=0.2) for s in range(6)]
    small = [synthetic.scattered_case(40 + s, n_z=101, n_y=5, jitter=0.3) for s in range(2)]
    ops.ResampleJob(small[:1], 8, 16).run_noalloc()        # first use on this device: symbol look-up, code already loaded
    keep = []
    for cases, (H, W) in ((big, (64, 512)), (small, (16, 40))):
        ref_job = ops.ResampleJob(cases, H, W)
        ref = ref_job.run().clone()
        ref_valid = ref_job.n_valid_slices()
        job = ops.ResampleJob(cases, H, W)
        job.out.fill_(float("nan"))
        valid = torch.full((len(cases),), -1, dtype=torch.int32, device="cuda")
        keep += [ref_job, ref, job, valid]
        torch.cuda.synchronize()
        n_alloc = torch.cuda.memory_stats().get("allocation.all.allocated", 0)    # torch's allocator: call count
        free_before = torch.cuda.mem_get_info()[0]                                # the device: any hipMalloc at all
        out = job.run_noalloc(valid)
        torch.cuda.synchronize()
        assert torch.cuda.memory_stats().get("allocation.all.allocated", 0) == n_alloc
        assert torch.cuda.mem_get_info()[0] == free_before
        assert torch.equal(out.view(torch.int64), ref.view(torch.int64))
        assert np.array_equal(valid.cpu().numpy(), ref_valid)
    # beyond the built-in scratch: refused with a clear error, nothing launched
    lib = _lib.load()
    n = 400_000
    assert lib.cfd_resample_workspace_bytes(1, 512, 101, n) > lib.cfd_resample_noalloc_max_bytes()
    pts = np.zeros((n, 5))
    pts[:, 1] = np.linspace(-2e-3, 2e-3, n)
    pts[:, 2] = np.linspace(0, 50e-3, n)
    too_big = ops.ResampleJob([pts], 64, 512)
    with pytest.raises(_lib.CfdError, match="scratch"):
        too_big.run_noalloc()
