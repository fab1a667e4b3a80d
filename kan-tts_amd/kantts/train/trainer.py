"""Trainers: the loop / logging / checkpoint shell around the MI355X training steps.

Mirrors the reference's ``Trainer``, ``Sambert_Trainer`` and ``GAN_Trainer`` (kantts/train/trainer.py:55-1043): same
constructor arguments, batch formats, checkpoint layout (``{"model", "optimizer", "scheduler", "steps"}``;
GAN: nested ``generator`` / ``discriminator`` dicts) and interval semantics, so that checkpoints move between the two
code bases and ``kantts.bin.train_*`` can drive either.  Differences, all on the host side of the hot path:
  * the losses of a step stay on the device; they are accumulated there and read back once per log interval (the
    reference calls ``.item()`` 9-15 times per step, a host sync each);
  * gradient clipping is folded into the fused Adam (``ArenaAdam.set_grad_clip``) when the optimizer supports it;
  * ``Sambert_Trainer`` can replay the whole step from a hipGraph once batch shapes repeat (``graph=True``);
  * TensorBoard is optional (a no-op writer is used when it is not installed);
  * evaluation / intermediate-result dumping are out of the hot path: ``eval_step`` computes the losses only.
"""
import logging
import os
import sys
from collections import defaultdict

import torch

from kantts.train.gan_step import gan_train_step

try:  # pragma: no cover - optional dependency
    from torch.utils.tensorboard import SummaryWriter
except Exception:  # noqa: BLE001
    SummaryWriter = None


class _NullWriter:
    def add_scalar(self, *a, **k):
        pass


def traversal_dict(d, func):
    for k, v in d.items():
        if isinstance(v, dict):
            traversal_dict(v, func)
        else:
            func(k, v)


def distributed_init():
    """env:// process-group init, one process per GPU (reference :25-52); RCCL is torch's "nccl" backend."""
    world_size = int(os.environ.get("WORLD_SIZE", 1))
    local_rank = int(os.environ.get("LOCAL_RANK", os.environ.get("RANK", 0)))
    if world_size > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        torch.cuda.set_device(local_rank)
        torch.distributed.init_process_group("nccl", init_method="env://")
    return world_size > 1, local_rank, world_size


class Trainer(object):
    def __init__(self, config, model, optimizer, scheduler, criterion, device, sampler, train_loader, valid_loader,
                 max_epochs=None, max_steps=None, save_dir=None, save_interval=1, valid_interval=1, log_interval=10,
                 grad_clip=None):
        self.model, self.optimizer, self.scheduler, self.criterion = model, optimizer, scheduler, criterion
        self.device, self.sampler = device, sampler
        self.train_loader, self.valid_loader = train_loader, valid_loader
        self.steps, self.epoch = 1, 0
        self.save_dir, self.save_interval, self.valid_interval = save_dir, save_interval, valid_interval
        self.log_interval, self.grad_clip, self.config = log_interval, grad_clip, config
        self.total_train_loss = defaultdict(float)
        self.total_eval_loss = defaultdict(float)
        self.distributed = config.get("distributed", False)
        self.rank = config.get("rank", 0)
        self.log_dir = self.ckpt_dir = None
        self.writer = _NullWriter()
        if save_dir is not None:
            self.log_dir, self.ckpt_dir = os.path.join(save_dir, "log"), os.path.join(save_dir, "ckpt")
            os.makedirs(self.log_dir, exist_ok=True)
            os.makedirs(self.ckpt_dir, exist_ok=True)
            if SummaryWriter is not None and self.rank == 0:
                self.writer = SummaryWriter(self.log_dir)
        self.max_epochs = sys.maxsize if max_epochs is None else int(max_epochs)
        self.max_steps = sys.maxsize if max_steps is None else int(max_steps)
        self.finish_training = False
        self._device_losses = {}  # name -> device scalar accumulated since the last log

    # ------------------------------------------------------------------ bookkeeping
    def _modules(self):
        out = []
        traversal_dict(self.model, lambda k, m: out.append(m)) if isinstance(self.model, dict) else out.append(self.model)
        return out

    def set_model_state(self, state="train"):
        if state not in ("train", "eval"):
            raise ValueError("state must be either 'train' or 'eval'.")
        for m in self._modules():
            m.train(state == "train")

    def _accumulate(self, prefix, losses):
        for k, v in losses.items():
            key = "%s/%s" % (prefix, k)
            # clone: under graph replay ``v`` is the graph's static loss buffer, which the next replay overwrites
            v = v.detach().clone() if torch.is_tensor(v) else torch.tensor(float(v), device=self.device)
            self._device_losses[key] = v if key not in self._device_losses else self._device_losses[key] + v

    def _flush_losses(self, into, prefix):
        """Move the accumulated ``prefix/*`` sums to the host (ONE sync); other prefixes keep accumulating, so an
        evaluation in the middle of a logging interval neither drains nor rescales the train sums."""
        keys = [k for k in self._device_losses if k.startswith(prefix + "/")]
        if keys:
            vals = torch.stack([self._device_losses[k].float().reshape(()) for k in keys]).cpu().tolist()
            for k, v in zip(keys, vals):
                into[k] += v
                del self._device_losses[k]

    def write_to_tensorboard(self, loss):
        for key, value in loss.items():
            self.writer.add_scalar(key, value, self.steps)

    def check_save_interval(self):
        if self.ckpt_dir is not None and self.steps % self.save_interval == 0:
            self.save_checkpoint(os.path.join(self.ckpt_dir, "checkpoint_{}.pth".format(self.steps)))
            logging.info("Checkpoint saved at step {}".format(self.steps))

    def check_log_interval(self):
        if self.steps % self.log_interval == 0:
            self._flush_losses(self.total_train_loss, "train")
            for key in self.total_train_loss:
                self.total_train_loss[key] /= self.config.get("log_interval_steps", self.log_interval)
                logging.info(f"(Steps: {self.steps}) {key} = {self.total_train_loss[key]:.4f}.")
            self.write_to_tensorboard(self.total_train_loss)
            self.total_train_loss = defaultdict(float)
            traversal_dict(self.scheduler, lambda key, sche: self.write_to_tensorboard(
                {"{}_lr".format(key): sche.get_last_lr()[0]}))

    def check_eval_interval(self):
        if self.valid_loader is not None and self.valid_interval > 0 and self.steps % self.valid_interval == 0:
            self.eval_epoch()

    def check_stop_training(self):
        if self.steps >= self.max_steps or self.epoch >= self.max_epochs:
            self.finish_training = True

    # ------------------------------------------------------------------ loops
    def train(self):
        self.set_model_state("train")
        while True:
            self.train_epoch()
            self.epoch += 1
            self.check_stop_training()
            if self.finish_training:
                break

    def train_epoch(self):
        for batch in self.train_loader:
            self.train_step(batch)
            if self.rank == 0:
                self.check_eval_interval()
                self.check_save_interval()
                self.check_log_interval()
            self.steps += 1
            self.check_stop_training()
            if self.finish_training:
                break
        logging.info("Epoch {} finished".format(self.epoch))
        if self.distributed and self.sampler and self.sampler.get("train") is not None:
            self.sampler["train"].set_epoch(self.epoch)

    @torch.no_grad()
    def eval_epoch(self):
        logging.info(f"(Epoch: {self.epoch}) Start evaluation.")
        self.set_model_state("eval")
        n = 0
        for n, batch in enumerate(self.valid_loader, 1):
            self.eval_step(batch)
        self._flush_losses(self.total_eval_loss, "eval")
        for key in self.total_eval_loss:
            self.total_eval_loss[key] /= max(n, 1)
            logging.info(f"(Steps: {self.steps}) {key} = {self.total_eval_loss[key]:.4f}.")
        self.write_to_tensorboard(self.total_eval_loss)
        self.total_eval_loss = defaultdict(float)
        self.set_model_state("train")

    def train_step(self, batch):
        raise NotImplementedError

    def eval_step(self, batch):
        raise NotImplementedError


class Sambert_Trainer(Trainer):
    """SAM-BERT trainer (reference :677-1043).  ``graph=True`` captures forward + losses + backward + clip + Adam
    into one hipGraph the first time a batch shape is seen and replays it for every later batch of that shape."""

    KEY = "KanTtsSAMBERT"

    def __init__(self, *args, graph=False, **kwargs):
        super().__init__(*args, **kwargs)
        params = self.config["Model"][self.KEY]["params"]
        self.with_MAS, self.fp_enable = params.get("MAS", False), params.get("FP", False)
        if self.fp_enable:
            raise NotImplementedError("filled-pause training is outside the hot path (DESIGN.md section 7)")
        # graph=True serves the MAS step too since round 6 (the CTC criterion is a device-side launch: csrc/ctc.hip)
        self.graph = graph
        self._graphs = {}
        self._ctl_group = None  # host-side (gloo) group of the data-parallel graph-cache agreement, made on first use
        self.max_graphs = 16  # captured steps kept (one per padded batch shape), LRU
        if self.grad_clip is not None and hasattr(self.optimizer[self.KEY], "set_grad_clip"):
            self.optimizer[self.KEY].set_grad_clip(self.grad_clip)

    def _to_device(self, batch):
        names = dict(inputs_ling="input_lings", inputs_emotion="input_emotions", inputs_speaker="input_speakers",
                     input_lengths="valid_input_lengths", output_lengths="valid_output_lengths",
                     mel_targets="mel_targets", duration_targets="durations", pitch_targets="pitch_contours",
                     energy_targets="energy_contours", attn_priors="attn_priors")
        return {k: (batch[v].to(self.device, non_blocking=True) if batch.get(v) is not None else None)
                for k, v in names.items()}

    def _losses(self, b, res):
        from kantts.train.loss import sambert_loss_sum

        total, losses = sambert_loss_sum(self.criterion["MelReconLoss"], self.criterion["ProsodyReconLoss"], b, res,
                                         prosody_lengths=res["valid_inter_lengths"])
        losses = dict(losses)
        if self.with_MAS:  # reference :871-884 / :970-983
            ctc = self.criterion["AttentionCTCLoss"](res["attn_logprob"], b["input_lengths"], b["output_lengths"])
            kl = self.criterion["AttentionBinarizationLoss"](self.epoch, res["attn_hard"], res["attn_soft"])
            total = total + ctc + kl
            losses.update(attn_ctc_loss=ctc, attn_kl_loss=kl)
        losses["TotalLoss"] = total
        return total, losses

    def train_step(self, batch):
        b = self._to_device(batch)
        net, opt, sch = self.model[self.KEY], self.optimizer[self.KEY], self.scheduler[self.KEY]
        if self.graph:
            # the attention band width of the batch, from the HOST copy (no device synchronisation) -- a batch assembled on
            # the device brings it along (DeviceAMSet.batch): the captured step has one shape for bands up to 16 (one launch
            # per decoder block) and another above
            from kantts.models.sambert.kantts_sambert import band_width_of

            bw = batch.get("band_width")
            if self.with_MAS:
                from kantts._hip import ops_bf16

                bw = ops_bf16.PB_MAX_BAND + 1  # the durations are aligned on the device: the wide-band form (graph_step.py)
            elif bw is None:
                bw = band_width_of(batch["durations"], batch["valid_input_lengths"], net.mel_decoder.r)
            return self._graph_step(b, bw)
        from kantts._hip import ops

        if b["mel_targets"].is_cuda:
            ops.advance_rng(self.device)
        res = net(**b)
        total, losses = self._losses(b, res)
        self._accumulate("train", losses)
        self._accumulate("train", {"batch_size": float(b["mel_targets"].size(0))})
        opt.zero_grad()
        total.backward()
        if self.grad_clip is not None and not hasattr(opt, "set_grad_clip"):
            torch.nn.utils.clip_grad_norm_(net.parameters(), self.grad_clip)
        opt.step()
        sch.step()
        return total

    def _agree_on_graph(self, shape_key, wide):
        """Data parallel: building a captured step issues collectives (warm-up steps all-reduce gradients, the capture form
        is agreed with a MIN all-reduce), replaying one issues others, and both the band class of a batch and the contents of
        the LRU cache are PER RANK.  So before the lookup the ranks agree -- one MAX all-reduce of three flags over a host
        (gloo) group, no device synchronisation: (this batch needs the wide-band form, no narrow-form graph for this shape
        here, no wide-form graph for this shape here).  Every rank then uses the wide form if any rank needs it, and every
        rank BUILDS if any rank has to (a rank that still had the graph rebuilds it: the collectives of a build have to
        pair up).  Single process: the local answer."""
        import torch.distributed as dist

        missing_narrow, missing_wide = (shape_key + (True,)) not in self._graphs, (shape_key + (False,)) not in self._graphs
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            if self._ctl_group is None:
                self._ctl_group = dist.group.WORLD if dist.get_backend() == "gloo" else dist.new_group(backend="gloo")
            flags = torch.tensor([int(wide), int(missing_narrow), int(missing_wide)], dtype=torch.int32)
            dist.all_reduce(flags, op=dist.ReduceOp.MAX, group=self._ctl_group)
            wide, missing_narrow, missing_wide = (bool(int(v)) for v in flags)
        return wide, (missing_wide if wide else missing_narrow)

    def _graph_step(self, b, band_width):
        from kantts._hip import ops_bf16
        from kantts.train.graph_step import GraphedSambertStep

        shape_key = tuple((k, tuple(v.shape)) for k, v in b.items() if v is not None)
        wide, build = self._agree_on_graph(shape_key, band_width > ops_bf16.PB_MAX_BAND)
        key = shape_key + (not wide,)
        g = self._graphs.pop(key, None)
        if build:
            g = None  # (data parallel: another rank has to build, so this one builds with it)
            if len(self._graphs) >= self.max_graphs:  # least recently used shape goes (dicts keep insertion order)
                self._graphs.pop(next(iter(self._graphs)))
            mas = ({k: self.criterion[k] for k in ("AttentionCTCLoss", "AttentionBinarizationLoss")} if self.with_MAS
                   else None)
            g = GraphedSambertStep(self.model[self.KEY], self.optimizer[self.KEY], self.scheduler[self.KEY],
                                   self.criterion["MelReconLoss"], self.criterion["ProsodyReconLoss"],
                                   {k: v for k, v in b.items() if v is not None}, band_width=band_width, force_wide=wide,
                                   mas_criteria=mas)
        else:
            g.load_batch({k: v for k, v in b.items() if v is not None}, band_width=band_width)
        self._graphs[key] = g  # most recently used last
        g.set_epoch(self.epoch)
        g()
        self._accumulate("train", {"TotalLoss": g.loss})
        return g.loss

    @torch.no_grad()
    def eval_step(self, batch):
        b = self._to_device(batch)
        res = self.model[self.KEY](**b)
        _, losses = self._losses(b, res)
        self._accumulate("eval", losses)

    def save_checkpoint(self, checkpoint_path):
        net = self.model[self.KEY]
        state_dict = {"optimizer": self.optimizer[self.KEY].state_dict(), "scheduler": self.scheduler[self.KEY].state_dict(),
                      "steps": self.steps, "model": getattr(net, "module", net).state_dict()}
        os.makedirs(os.path.dirname(checkpoint_path), exist_ok=True)
        torch.save(state_dict, checkpoint_path)

    def load_checkpoint(self, checkpoint_path, restore_training_state=False, strict=True):
        state_dict = torch.load(checkpoint_path, map_location="cpu")
        net = self.model[self.KEY]
        getattr(net, "module", net).load_state_dict(state_dict["model"], strict=strict)
        if restore_training_state:
            if "optimizer" in state_dict:
                self.optimizer[self.KEY].load_state_dict(state_dict["optimizer"])
            if "scheduler" in state_dict:
                self.scheduler[self.KEY].load_state_dict(state_dict["scheduler"])
            if "steps" in state_dict:
                self.steps = state_dict["steps"]


class GAN_Trainer(Trainer):
    """HiFi-GAN trainer (reference :276-675); batch = (y wav (B,1,T), x mel (B,C,T/hop))."""

    def __init__(self, *args, graph=False, **kwargs):
        """``graph=True``: once both phases are active, the whole step (both phases, three Adam updates) is captured into
        one hipGraph per batch shape and replayed (kantts/train/gan_graph_step.py); the vocoder dataset crops every batch
        to one shape, so this is a single capture.  Needs the arena optimizers and a single process."""
        super().__init__(*args, **kwargs)
        self.graph = graph
        self._graphs = {}

    def _graph_ready(self):
        return (self.graph and self.steps > self.config.get("discriminator_train_start_steps", 0)
                and self.steps >= self.config.get("generator_train_start_steps", 0))

    def train_step(self, batch):
        y, x = batch
        y, x = y.to(self.device, non_blocking=True), x.to(self.device, non_blocking=True)
        g = None
        if self._graph_ready():
            from kantts.train.gan_graph_step import GraphedGanStep

            key = (tuple(y.shape), tuple(x.shape))
            g = self._graphs.get(key)
            if g is None:
                if len(self._graphs) >= 4:
                    self._graphs.pop(next(iter(self._graphs)))
                try:
                    g = self._graphs[key] = GraphedGanStep(self.model, self.optimizer, self.scheduler, self.criterion,
                                                           self.config, y, x, steps=self.steps)
                except (NotImplementedError, ValueError) as exc:
                    # NSF generators (host-seeded excitation), non-arena optimizers, a capture the runtime refused
                    # (gan_graph_step.CaptureRefused; data-parallel: on any rank, agreed across ranks): the step cannot be
                    # captured.  Any other error (out of memory, a kernel's failed check in the warm-up steps) propagates.
                    # Say so once and keep training with the eager step (the run must not die thousands of steps in,
                    # when both phases first become active).
                    logging.warning("[GAN_Trainer] capture_step is off for this run: %s", exc)
                    self.graph = False
                    g = None
            else:
                g.load_batch(y, x)
        if g is not None:
            losses = g()
        else:
            losses = gan_train_step(self.model, self.optimizer, self.scheduler, self.criterion, self.config, y, x,
                                    steps=self.steps)
        self._accumulate("train", losses)
        return losses

    @torch.no_grad()
    def eval_step(self, batch):
        from kantts.train.gan_step import discriminator_loss, generator_loss

        y, x = batch
        y, x = y.to(self.device), x.to(self.device)
        _, gl, _ = generator_loss(self.model, self.criterion, x, y)
        _, dl = discriminator_loss(self.model, self.criterion, x, y)
        self._accumulate("eval", {**gl, **dl})

    def save_checkpoint(self, checkpoint_path):
        def sd(m):
            return getattr(m, "module", m).state_dict()

        dis = self.model["discriminator"]
        state_dict = {
            "optimizer": {"generator": self.optimizer["generator"].state_dict(),
                          "discriminator": {k: o.state_dict() for k, o in self.optimizer["discriminator"].items()}},
            "scheduler": {"generator": self.scheduler["generator"].state_dict(),
                          "discriminator": {k: s.state_dict() for k, s in self.scheduler["discriminator"].items()}},
            "steps": self.steps,
            "model": {"generator": sd(self.model["generator"]), "discriminator": {k: sd(m) for k, m in dis.items()}},
        }
        os.makedirs(os.path.dirname(checkpoint_path), exist_ok=True)
        torch.save(state_dict, checkpoint_path)

    def load_checkpoint(self, checkpoint_path, restore_training_state=False, strict=True):
        state_dict = torch.load(checkpoint_path, map_location="cpu")

        def ld(m, s):
            getattr(m, "module", m).load_state_dict(s, strict=strict)

        ld(self.model["generator"], state_dict["model"]["generator"])
        for name, s in state_dict["model"]["discriminator"].items():
            ld(self.model["discriminator"][name], s)
        if restore_training_state:
            self.steps = state_dict.get("steps", self.steps)
            for part in ("optimizer", "scheduler"):
                if part in state_dict:
                    getattr(self, part)["generator"].load_state_dict(state_dict[part]["generator"])
                    for name, s in state_dict[part]["discriminator"].items():
                        getattr(self, part)["discriminator"][name].load_state_dict(s)
