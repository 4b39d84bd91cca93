"""csrc/sort.hip: the hand-written stable LSD radix sort of (u64 key, u32 value) pairs behind the descriptor index build (csrc/fmatch.hip)
against numpy's stable argsort - every size class (one workgroup, a ragged last workgroup, hundreds of workgroups), every pass count
(1 to 64 bits), keys with many duplicates (stability is what the index build relies on: rows enter in index order and ties keep it)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(ctx, keys, vals, end_bit):
    dev = torch.device("cuda", 0)
    n = len(keys)
    d_k = torch.from_numpy(keys.view(np.int64)).to(dev); d_v = torch.from_numpy(vals.view(np.int32)).to(dev)
    o_k = torch.full((max(n, 1),), -1, dtype=torch.int64, device=dev); o_v = torch.full((max(n, 1),), -1, dtype=torch.int32, device=dev)
    ctx.radix_sort_pairs_dev(d_k.data_ptr() if n else None, o_k.data_ptr(), d_v.data_ptr() if n else None, o_v.data_ptr(), n, end_bit)
    assert d_k.cpu().numpy().view(np.uint64).tobytes() == keys.tobytes()          # the input is left alone
    return o_k[:n].cpu().numpy().view(np.uint64), o_v[:n].cpu().numpy().view(np.uint32)


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 1000, 2047, 2048, 2049, 8192, 150001, 1 << 20, 3000001])   # (above 2 M pairs: the ticketed table scan)
@pytest.mark.parametrize("end_bit", [1, 7, 8, 9, 33, 42, 64])
def test_radix_sort_equals_stable_argsort(ctx, n, end_bit):
    rng = np.random.default_rng(n * 131 + end_bit)
    # high bits random (must be ignored above end_bit), low bits from a small alphabet: many duplicates
    keys = rng.integers(0, 1 << 63, n, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, n, dtype=np.uint64)
    if end_bit > 16:
        dup = rng.random(n) < 0.5
        keys[dup] = (keys[dup] & ~np.uint64((1 << 16) - 1)) | np.uint64(rng.integers(0, 7))
    vals = np.arange(n, dtype=np.uint32)[::-1].copy()
    k, v = _run(ctx, keys, vals, end_bit)
    low = keys & np.uint64((1 << end_bit) - 1) if end_bit < 64 else keys
    order = np.argsort(low, kind="stable")
    assert k.tobytes() == keys[order].tobytes() and v.tobytes() == vals[order].tobytes()


def test_radix_sort_degenerate_inputs(ctx):
    for keys in (np.zeros(5000, np.uint64), np.full(5000, np.uint64(2 ** 64 - 1)), np.arange(5000, dtype=np.uint64)[::-1].copy(), np.arange(70000, dtype=np.uint64) % np.uint64(3)):
        vals = np.arange(len(keys), dtype=np.uint32)
        k, v = _run(ctx, keys, vals, 64)
        order = np.argsort(keys, kind="stable")
        assert k.tobytes() == keys[order].tobytes() and v.tobytes() == vals[order].tobytes()
    k, v = _run(ctx, np.zeros(0, np.uint64), np.zeros(0, np.uint32), 33)
    assert len(k) == 0
