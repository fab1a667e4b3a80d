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
        assert isinstance(got.pop("band_width"), int)  # the batch's attention band width rides along (host integer)
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


def test_pinned_prefetcher_carries_the_band_width_of_acoustic_batches():
    """with ``r`` the prefetcher computes x_band_width from the HOST batch and hands it on as the host integer
    ``band_width`` (what DeviceAMSet.batch does), so that the trainer's captured step does not read it back from the
    device; it equals the model's band_width_of; vocoder batches (tuples) and dicts without durations pass unchanged;
    short vocoder items are refused when the device set is built, not mid-epoch."""
    import numpy as np

    from kantts.datasets.device_batching import DeviceVocSet, PinnedPrefetcher
    from kantts.models.sambert.kantts_sambert import band_width_of

    g = torch.Generator().manual_seed(0)
    batches = []
    for _ in range(3):
        d = torch.randint(1, 60, (4, 9), generator=g)
        n = torch.tensor([9, 4, 7, 1])
        batches.append({"durations": d, "valid_input_lengths": n, "mel_targets": torch.randn(4, 5, 2, generator=g)})
    out = list(PinnedPrefetcher(batches, "cpu", r=3))
    for a, b in zip(batches, out):
        assert b["band_width"] == band_width_of(a["durations"], a["valid_input_lengths"], 3)
        assert isinstance(b["band_width"], int) and torch.equal(a["mel_targets"], b["mel_targets"])
    assert "band_width" not in list(PinnedPrefetcher(batches, "cpu"))[0]
    assert "band_width" not in list(PinnedPrefetcher([{"x": torch.zeros(2)}], "cpu", r=3))[0]
    with pytest.raises(ValueError, match="a crop needs more than"):
        DeviceVocSet([(np.zeros(4 * 8, np.float32), np.zeros((4, 3), np.float32))], 8, 32, "cpu")


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


def _trainer_items(n=8, seed=4):
    """Duration-supervised items with ids inside the PinYin vocabularies of the tiny SAM-BERT config."""
    rng = np.random.RandomState(seed)
    hi = (140, 7, 5, 5, 30, 1)
    items = []
    for _ in range(n):
        nsym = int(rng.randint(6, 13))  # incl. the trailing "~"
        dur = rng.randint(1, 6, size=nsym - 1).astype(np.int64)
        ling = [rng.randint(0, hi[k], size=nsym).astype(np.int64) for k in range(6)]
        items.append((ling, rng.randn(int(dur.sum()), 80).astype(np.float32), dur, rng.randn(nsym).astype(np.float32),
                      rng.randn(nsym).astype(np.float32), None, None, None))
    return items


def _two_trainer_steps(device, feed):
    """Two Sambert_Trainer steps on batches assembled by the host collate ("host") or on the device ("device")."""
    from test_trainer import _sambert_setup

    from kantts.datasets.batching import am_collate
    from kantts.datasets.device_batching import DeviceAMSet, DeviceCorpusLoader

    items = _trainer_items()
    pad_ids = [146, 9, 7, 7, 35, 3]
    index_batches = [[0, 3, 5], [7, 1, 2]]
    tr, _ = _sambert_setup(device, "/tmp/kantts_feeder_test", seed=0)
    if feed == "device":
        loader = DeviceCorpusLoader(DeviceAMSet(items, 3, pad_ids, device), 3, batches=index_batches)
        assert len(loader) == 2
    else:
        loader = [am_collate([items[i] for i in idx], 3, pad_ids) for idx in index_batches]
    losses = []
    for batch in loader:
        if feed == "device":
            assert isinstance(batch["band_width"], int) and batch["mel_targets"].device.type == torch.device(device).type
        losses.append(float(tr.train_step({k: v for k, v in batch.items() if k != "band_width"} if feed == "host" else batch)))
        tr.steps += 1
    flat = torch.cat([p.detach().reshape(-1).cpu() for p in tr.model["KanTtsSAMBERT"].parameters()])
    return losses, flat


def _feeder_check(device):
    la, wa = _two_trainer_steps(device, "host")
    lb, wb = _two_trainer_steps(device, "device")
    if device == "cpu":
        assert la == lb and torch.equal(wa, wb)  # identical batches -> identical steps
    else:
        # on the device the weight gradients are fp32 atomics whose order differs from run to run: identical up to the
        # noise floor of two host-fed runs
        lc, wc = _two_trainer_steps(device, "host")
        floor = float((wa - wc).norm() / wa.norm())
        assert float((wa - wb).norm() / wa.norm()) <= max(3 * floor, 1e-6), floor
        assert all(abs(x - y) <= 1e-5 * max(1.0, abs(x)) for x, y in zip(la, lb))
    from kantts.datasets.device_batching import DeviceAMSet
    from kantts.models.sambert.kantts_sambert import band_width_of

    items = _trainer_items()
    ds = DeviceAMSet(items, 3, [146, 9, 7, 7, 35, 3], device)
    for idx in ([0, 3, 5], [7, 1, 2], [4]):
        b = ds.batch(idx)
        assert b["band_width"] == band_width_of(b["durations"].cpu(), b["valid_input_lengths"].cpu(), 3)


def test_trainer_steps_fed_from_the_device_corpus_equal_host_fed_steps_emulated():
    with emulation():
        _feeder_check("cpu")


@pytest.mark.gpu
def test_trainer_steps_fed_from_the_device_corpus_equal_host_fed_steps_gpu():
    _feeder_check("cuda")
