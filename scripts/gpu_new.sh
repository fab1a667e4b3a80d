#!/bin/bash
# Pre-check of the GPU tests added last (before spending a full validation visit).
cd "$(dirname "$0")/.."
timeout 300 python -m pytest tests/test_trainer.py -m gpu -q --timeout=300 -p no:cacheprovider -k "curve" 2>&1 | tail -12
