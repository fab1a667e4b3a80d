"""Batch assembly (batching.py, device_batching.py) and the acoustic-model / vocoder datasets (dataset.py)."""
