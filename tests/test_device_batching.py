"""Device-side batch assembly (kantts/datasets/device_batching.py on csrc/batching.hip) against the host
collate functions, which are themselves pinned bit-exactly to the reference's (tests/test_batching.py, tests/golden/
collate.pt): same seeded crop starts, same pad ids, same duration parking -- identical tensors."""
import numpy as np
import pytest
import torch

from util import emulation


def _voc_items(n=7, hop=16, seed=0):
    rng = np.random.RandomState(seed)
    items = []
    for _ in range(n):
        frames = int(rng.randint(40, 90))
        items.append((rng.randn(frames * hop).astype(np.float32), rng.randn(frames, 20).astype(np.float32)))
    return items


def _am_items(n=6, r=3, seed=1):
    rng = np.random.RandomState(seed)
    items = []
    for _ in range(n):
        nsym = int(rng.randint(5, 15))  # incl. the trailing "~"
        dur = rng.randint(1, 6, size=nsym - 1).astype(np.int64)
        frames = int(dur.sum())
        ling = [rng.randint(0, 9, size=nsym).astype(np.int64) for _ in range(6)]
        items.append((ling, rng.randn(frames, 80).astype(np.float32), dur, rng.randn(nsym).astype(np.float32),
                      rng.randn(nsym).astype(np.float32), None, None, None))
    return items


def _check_voc(device):
    from kantts.datasets.batching import voc_collate
    from kantts.datasets.device_batching import DeviceVocSet

    items = _voc_items()
    ds = DeviceVocSet(items, hop_length=16, batch_max_steps=320, device=device)
    for seed, idx in ((3, [0, 2, 5]), (4, [6, 1, 1, 3, 4])):
        w_ref, m_ref = voc_collate([items[i] for i in idx], 16, 320, rng=np.random.RandomState(seed))
        w, m = ds.batch(idx, rng=np.random.RandomState(seed))
        assert w.shape == w_ref.shape and m.shape == m_ref.shape
        assert torch.equal(w.cpu(), w_ref) and torch.equal(m.cpu(), m_ref)


def _check_am(device):
    from kantts.datasets.batching import am_collate
    from kantts.datasets.device_batching import DeviceAMSet

    items = _am_items()
    pad_ids = [11, 12, 13, 14, 15, 16]
    ds = DeviceAMSet(items, r=3, pad_ids=pad_ids, device=device)
    for idx in ([0, 1, 2], [5, 3, 3, 4], [2]):
        ref = am_collate([items[i] for i in idx], 3, pad_ids)
        got = ds.batch(idx)
        assert set(got) == set(ref)
        for k, v in ref.items():
            if v is None:
                assert got[k] is None, k
            else:
                assert got[k].shape == v.shape and got[k].dtype == v.dtype, k
                assert torch.equal(got[k].cpu(), v), k


def test_device_voc_batches_equal_host_collate_emulated():
    with emulation():
        _check_voc("cpu")


def test_device_am_batches_equal_host_collate_emulated():
    with emulation():
        _check_am("cpu")


def test_pinned_prefetcher_passes_batches_through_on_cpu():
    from kantts.datasets.device_batching import PinnedPrefetcher

    batches = [{"a": torch.full((2, 3), float(i)), "n": None, "t": (torch.arange(4) + i, "tag")} for i in range(4)]
    out = list(PinnedPrefetcher(batches, "cpu"))
    assert len(out) == 4
    for i, b in enumerate(out):
        assert torch.equal(b["a"], batches[i]["a"]) and b["n"] is None and b["t"][1] == "tag"
        assert torch.equal(b["t"][0], batches[i]["t"][0])


@pytest.mark.gpu
def test_device_batches_equal_host_collate_gpu():
    _check_voc("cuda")
    _check_am("cuda")


@pytest.mark.gpu
def test_pinned_prefetcher_overlaps_and_preserves_order_gpu():
    from kantts.datasets.device_batching import PinnedPrefetcher

    g = torch.Generator().manual_seed(0)
    batches = [(torch.randn(32, 1, 8192, generator=g), torch.randn(32, 80, 32, generator=g)) for _ in range(6)]
    acc = []
    for y, x in PinnedPrefetcher(batches, "cuda"):
        assert y.is_cuda and x.is_cuda
        acc.append((y.sum() + x.sum()).item())  # consumer work on the current stream
    for a, (y, x) in zip(acc, batches):
        assert abs(a - float(y.sum() + x.sum())) < 1e-2 * max(1.0, abs(a))
