"""Offline feature extraction that runs on the device (audio_processor)."""
