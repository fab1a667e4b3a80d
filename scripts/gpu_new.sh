#!/bin/bash
cd "$(dirname "$0")/.."
timeout 300 python -m pytest tests/test_hifigan_nsf.py tests/test_sambert_se.py tests/test_audio_processor.py -m gpu -q --timeout=300 -p no:cacheprovider 2>&1 | tail -15
