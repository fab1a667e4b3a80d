"""Shared pieces of the training entry points (config merge, devices)."""
import logging
import os
import time

import torch
import yaml

logging.basicConfig(format="%(asctime)s %(levelname)-4s [%(filename)s:%(lineno)d] %(message)s",
                    datefmt="%Y-%m-%d:%H:%M:%S", level=logging.INFO)


def count_parameters(model):
    return sum(p.numel() for p in model.parameters() if p.requires_grad)


def load_config(model_config, root_dir):
    """audio_config.yaml of the first data root (when present) updated with the model config (reference
    train_sambert.py:62-67)."""
    config = {}
    if root_dir:
        audio_config = os.path.join(root_dir[0], "audio_config.yaml")
        if os.path.exists(audio_config):
            with open(audio_config) as f:
                config = yaml.load(f, Loader=yaml.Loader)
    with open(model_config) as f:
        config.update(yaml.load(f, Loader=yaml.Loader))
    config["create_time"] = time.strftime("%Y-%m-%d %H:%M:%S", time.localtime())
    return config


def setup_device():
    """(distributed, device, local_rank, world_size): one process per GPU, env:// rendezvous."""
    from kantts.train.trainer import distributed_init

    if not torch.cuda.is_available():
        return False, torch.device("cpu"), 0, 1
    distributed, local_rank, world_size = distributed_init()
    torch.cuda.set_device(local_rank)
    return distributed, torch.device("cuda", local_rank), local_rank, world_size


def dataset_module():
    try:
        import kantts.datasets.dataset as ds  # the reference's data pipeline: not part of the hot path

        return ds
    except ImportError:
        return None
