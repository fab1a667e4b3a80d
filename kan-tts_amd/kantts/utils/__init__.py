"""audio_torch (mel-STFT on the HIP kernels), ling_unit (symbol tables of the acoustic model), synthetic workloads."""
