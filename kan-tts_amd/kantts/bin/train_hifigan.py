"""HiFi-GAN training entry point (reference kantts/bin/train_hifigan.py:33-215): same flags and stage-directory
layout.  ``--synthetic N`` trains on N seeded (wav, mel) batches when kantts.datasets is not installed."""
import argparse
import logging
import os
import sys

import torch
import yaml

from kantts.bin._common import count_parameters, dataset_module, load_config, setup_device
from kantts.models import model_builder
from kantts.train.loss import criterion_builder
from kantts.train.trainer import GAN_Trainer


def train(model_config, root_dir, stage_dir, resume_path=None, local_rank=0, synthetic=0, device_corpus="off"):
    distributed, device, local_rank, world_size = setup_device()
    if local_rank != 0:
        sys.stdout = open(os.devnull, "w")
        logging.getLogger().disabled = True
    root_dir = root_dir if isinstance(root_dir, list) else [root_dir]
    if local_rank == 0:
        os.makedirs(stage_dir, exist_ok=True)
    config = load_config(model_config, root_dir) if not isinstance(model_config, dict) else dict(model_config)
    if local_rank == 0:
        with open(os.path.join(stage_dir, "config.yaml"), "w") as f:
            yaml.dump(config, f, Dumper=yaml.Dumper, default_flow_style=None)
    if distributed:
        config["rank"], config["distributed"] = torch.distributed.get_rank(), True
    sampler = {"train": None, "valid": None}
    ds = dataset_module()
    if synthetic:
        hop = config.get("audio_config", {}).get("hop_length", 256)
        seg = config.get("batch_max_steps", 8192) // hop * hop
        B = config.get("batch_size", 16)
        g = torch.Generator().manual_seed(4321 + config.get("rank", 0))
        nsf = config["Model"]["Generator"]["params"].get("nsf_params") is not None

        def feats():
            mel = torch.randn(B, 80, seg // hop, generator=g)
            if not nsf:
                return mel
            # NSF generators read two more channels: f0 in Hz (0 where unvoiced) and the voiced flag
            uv = (torch.rand(B, 1, seg // hop, generator=g) > 0.3).float()
            f0 = (80.0 + 300.0 * torch.rand(B, 1, seg // hop, generator=g)) * uv
            return torch.cat([mel, f0, uv], dim=1)

        train_loader = [(torch.randn(B, 1, seg, generator=g).clamp(-1, 1), feats()) for _ in range(synthetic)]
        valid_loader = None
    elif ds is not None:
        from torch.utils.data import DataLoader

        train_set, valid_set = ds.get_voc_datasets(config, root_dir)
        if distributed:
            from torch.utils.data.distributed import DistributedSampler

            sampler["train"] = DistributedSampler(train_set, num_replicas=world_size, shuffle=True)
            sampler["valid"] = DistributedSampler(valid_set, num_replicas=world_size, shuffle=False)
        kw = dict(batch_size=config["batch_size"], num_workers=config["num_workers"], pin_memory=config["pin_memory"])
        train_loader = DataLoader(train_set, shuffle=not distributed, collate_fn=train_set.collate_fn,
                                  sampler=sampler["train"], **kw)
        valid_loader = DataLoader(valid_set, shuffle=not distributed, collate_fn=valid_set.collate_fn,
                                  sampler=sampler["valid"], **kw)
        if device_corpus != "off":  # the training set resident in HBM / staged through pinned memory (row f2 of SURVEY 8)
            from kantts.datasets.device_batching import make_train_loader

            train_loader = make_train_loader("voc", train_set, train_loader, device, config["batch_size"],
                                             sampler=sampler["train"], mode=device_corpus)
    else:
        raise ImportError("kantts.datasets is not installed (the data pipeline is outside this package): "
                          "pass --synthetic N or install the reference's dataset module")
    model, optimizer, scheduler = model_builder(config, device, local_rank, distributed)
    criterion = criterion_builder(config, device)
    logging.info("Generator parameters count: %d", count_parameters(model["generator"]))
    trainer = GAN_Trainer(config=config, model=model, optimizer=optimizer, scheduler=scheduler, criterion=criterion,
                          device=device, sampler=sampler, train_loader=train_loader, valid_loader=valid_loader,
                          max_steps=config.get("train_max_steps"), max_epochs=1 if synthetic else None,
                          save_dir=stage_dir, save_interval=config.get("save_interval_steps", 10 ** 9),
                          valid_interval=config.get("eval_interval_steps", 10 ** 9),
                          log_interval=config.get("log_interval_steps", 10),
                          # ``capture_step: true`` (or KANTTS_GAN_GRAPH=1): replay the step from one hipGraph; data-parallel
                          # replicas replay a chain of graph segments with the bucketed all-reduces between them
                          # (kantts/train/segments.py)
                          graph=(bool(config.get("capture_step", os.environ.get("KANTTS_GAN_GRAPH", "") == "1"))
                                 and torch.device(device).type == "cuda"))
    if resume_path is not None:
        trainer.load_checkpoint(resume_path, True, False)
        logging.info("Successfully resumed from %s.", resume_path)
    try:
        trainer.train()
    except (Exception, KeyboardInterrupt) as e:  # noqa: BLE001
        logging.error(e, exc_info=True)
        trainer.save_checkpoint(os.path.join(stage_dir, "ckpt", "checkpoint-%d.pth" % trainer.steps))
        raise
    return trainer


if __name__ == "__main__":
    parser = argparse.ArgumentParser(description="Train a model for speech synthesis")
    parser.add_argument("--model_config", type=str, required=True, help="model config file")
    parser.add_argument("--root_dir", nargs="+", type=str, default=[], help="root dir of dataset(s)")
    parser.add_argument("--stage_dir", type=str, required=True, help="stage dir of checkpoint, log and intermediate results")
    parser.add_argument("--resume_path", type=str, default=None, help="path to resume checkpoint")
    parser.add_argument("--local_rank", type=int, default=0, help="local rank for distributed training")
    parser.add_argument("--synthetic", type=int, default=0, help="train on N seeded synthetic batches (no dataset)")
    parser.add_argument("--device_corpus", default="off", choices=["off", "auto", "hbm", "pinned"],
                        help="training set resident in HBM with batches cropped on the device (hbm), host batches staged "
                             "through pinned memory one step ahead (pinned), hbm when it fits else pinned (auto), or the "
                             "reference's DataLoader as is (off)")
    a = parser.parse_args()
    train(a.model_config, a.root_dir, a.stage_dir, a.resume_path, a.local_rank, a.synthetic, a.device_corpus)
