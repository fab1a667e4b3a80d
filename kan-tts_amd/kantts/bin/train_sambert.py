"""SAM-BERT training entry point (reference kantts/bin/train_sambert.py:33-241): same flags, config handling,
stage-directory layout (config.yaml, ckpt/, log/), resume semantics and emergency checkpoint.  The data pipeline
(kantts.datasets) is outside the hot path: when it is not installed, ``--synthetic N`` trains on N seeded synthetic
batches of the config's shape (the benchmark workload), which is also how the step is smoke-tested end to end."""
import argparse
import logging
import os
import sys

import torch
import yaml

from kantts.bin._common import count_parameters, dataset_module, load_config, setup_device
from kantts.models import model_builder
from kantts.train.loss import criterion_builder
from kantts.train.trainer import Sambert_Trainer


def train(model_config, root_dir, stage_dir, resume_path=None, resume_bert_path=None, local_rank=0, synthetic=0,
          graph=False, device_corpus="off"):
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
        from kantts.utils.synthetic import SAMBERT_VOCAB, sambert_batch, sambert_mas_batch, to_collate_format

        params = config["Model"]["KanTtsSAMBERT"]["params"]
        for k, v in SAMBERT_VOCAB.items():
            params.setdefault(k, v)
        rank = config.get("rank", 0)
        make = sambert_mas_batch if params.get("MAS", False) else sambert_batch  # MAS: no durations, priors instead
        train_loader = [to_collate_format(make(B=config["batch_size"], seed=1234 + 1000 * rank + s))
                        for s in range(synthetic)]
        valid_loader = None
    elif ds is not None:
        from torch.utils.data import DataLoader

        meta = [os.path.join(d, "raw_metafile.txt") for d in root_dir]
        train_set, valid_set = ds.get_am_datasets(meta, root_dir, config, config["allow_cache"], split_ratio=0.98)
        if distributed:
            from torch.utils.data.distributed import DistributedSampler

            sampler["train"] = DistributedSampler(train_set, num_replicas=world_size, shuffle=True)
            sampler["valid"] = DistributedSampler(valid_set, num_replicas=world_size, shuffle=False)
        kw = dict(batch_size=config["batch_size"], num_workers=config["num_workers"], pin_memory=config["pin_memory"])
        train_loader = DataLoader(train_set, shuffle=not distributed, collate_fn=train_set.collate_fn,
                                  sampler=sampler["train"], **kw)
        valid_loader = DataLoader(valid_set, shuffle=not distributed, collate_fn=valid_set.collate_fn,
                                  sampler=sampler["valid"], **kw)
        config["Model"]["KanTtsSAMBERT"]["params"].update(train_set.ling_unit.get_unit_size())
        if device_corpus != "off":  # the training set resident in HBM / staged through pinned memory (row f2 of SURVEY 8)
            from kantts.datasets.device_batching import make_train_loader

            train_loader = make_train_loader("am", train_set, train_loader, device, config["batch_size"],
                                             sampler=sampler["train"], mode=device_corpus)
    else:
        raise ImportError("kantts.datasets could not be imported: pass --synthetic N")
    model, optimizer, scheduler = model_builder(config, device, local_rank, distributed)
    criterion = criterion_builder(config, device) if "Loss" in config else None
    if not criterion:
        from kantts.train.loss import MelReconLoss, ProsodyReconLoss

        criterion = {"MelReconLoss": MelReconLoss(), "ProsodyReconLoss": ProsodyReconLoss()}
    logging.info("Sambert model parameters count: %d", count_parameters(model["KanTtsSAMBERT"]))
    trainer = Sambert_Trainer(config=config, model=model, optimizer=optimizer, scheduler=scheduler,
                              criterion=criterion, device=device, sampler=sampler, train_loader=train_loader,
                              valid_loader=valid_loader, max_steps=config.get("train_max_steps"),
                              max_epochs=1 if synthetic else None, save_dir=stage_dir,
                              save_interval=config.get("save_interval_steps", 10 ** 9),
                              valid_interval=config.get("eval_interval_steps", 10 ** 9),
                              log_interval=config.get("log_interval_steps", 10), grad_clip=config.get("grad_norm"),
                              graph=graph)
    if resume_path is not None:
        trainer.load_checkpoint(resume_path, True, False)
        logging.info("Successfully resumed from %s.", resume_path)
    if resume_bert_path is not None:
        trainer.load_checkpoint(resume_bert_path, False, False)
    try:
        trainer.train()
    except (Exception, KeyboardInterrupt) as e:  # noqa: BLE001 - the reference saves an emergency checkpoint here too
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
    parser.add_argument("--resume_bert_path", type=str, default=None, help="path to resume pre-trained bert")
    parser.add_argument("--local_rank", type=int, default=0, help="local rank for distributed training")
    parser.add_argument("--synthetic", type=int, default=0, help="train on N seeded synthetic batches (no dataset)")
    parser.add_argument("--graph", action="store_true", help="replay the training step from a hipGraph")
    parser.add_argument("--device_corpus", default="off", choices=["off", "auto", "hbm", "pinned"],
                        help="training set resident in HBM with batches assembled on the device (hbm), host batches staged "
                             "through pinned memory one step ahead (pinned), hbm when it fits else pinned (auto), or the "
                             "reference's DataLoader as is (off)")
    a = parser.parse_args()
    train(a.model_config, a.root_dir, a.stage_dir, a.resume_path, a.resume_bert_path, a.local_rank, a.synthetic, a.graph, a.device_corpus)
