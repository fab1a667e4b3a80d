"""Model / optimizer / scheduler builders (reference kantts/models/__init__.py:16-164).

Same entry point: ``model_builder(config, device="cpu", rank=0, distributed=False)`` returning the
(model, optimizer, scheduler) dicts with the reference's keys.  On a HIP device the SAM-BERT
parameters are laid out in one flat arena so that the optimizer step (global-norm clip + Adam) is
two kernel launches and data-parallel gradient exchange is one RCCL all-reduce per arena slice
(kantts.train.optim); on CPU the plain torch optimizer of the reference is built (modules can be
constructed / checkpointed on the host, but have no CPU forward).
"""
import torch
from torch.nn.parallel import DistributedDataParallel

import kantts
import kantts.train.scheduler
from kantts.models.sambert.kantts_sambert import KanTtsSAMBERT, KanTtsTextsyBERT  # NOQA
from kantts.train.optim import ArenaAdam, ParamArena


def optimizer_builder(model_params, opt_name, opt_params):
    opt_cls = getattr(torch.optim, opt_name)
    return opt_cls(model_params, **opt_params)


def scheduler_builder(optimizer, sche_name, sche_params):
    scheduler_cls = getattr(kantts.train.scheduler, sche_name)
    return scheduler_cls(optimizer, **sche_params)


def _is_hip(device):
    return torch.device(device).type == "cuda"


def sambert_model_builder(config, device, rank, distributed, use_arena=None):
    model, optimizer, scheduler = {}, {}, {}
    mconf = config["Model"]["KanTtsSAMBERT"]
    net = KanTtsSAMBERT(mconf["params"]).to(device)
    model["KanTtsSAMBERT"] = net
    opt_type = mconf["optimizer"].get("type", "Adam")
    opt_params = mconf["optimizer"].get("params", {})
    if use_arena is None:
        use_arena = _is_hip(device) and opt_type == "Adam" and not opt_params.get("amsgrad", False)
    if use_arena:
        arena = ParamArena(net, bf16_shadow=True)  # operand images of the bf16-mode contractions
        optimizer["KanTtsSAMBERT"] = ArenaAdam(arena, **opt_params)
        if distributed:
            arena.enable_data_parallel()
    else:
        optimizer["KanTtsSAMBERT"] = optimizer_builder(net.parameters(), opt_type, opt_params)
        if distributed:
            model["KanTtsSAMBERT"] = DistributedDataParallel(net, device_ids=[rank], output_device=rank)
    scheduler["KanTtsSAMBERT"] = scheduler_builder(
        optimizer["KanTtsSAMBERT"], mconf["scheduler"].get("type", "StepLR"), mconf["scheduler"].get("params", {}))
    return model, optimizer, scheduler


def hifigan_model_builder(config, device, rank, distributed, use_arena=None):
    from kantts.models.hifigan import hifigan as _h

    model, optimizer, scheduler = {}, {}, {}
    model["discriminator"], optimizer["discriminator"], scheduler["discriminator"] = {}, {}, {}
    arena_nets = []

    def _opt(net, conf):
        otype, oparams = conf["optimizer"].get("type", "Adam"), conf["optimizer"].get("params", {})
        if (use_arena if use_arena is not None else _is_hip(device)) and otype == "Adam" and not oparams.get("amsgrad", False):
            arena = ParamArena(net)
            arena.build_weight_norm_images()
            if distributed:
                # three independent reducers (generator, MPD, MSD), each exchanging its gradient arena in a few large
                # bucketed all-reduces overlapped with its own backward; a discriminator arena is only armed by its own
                # zero_grad, i.e. in the discriminator phase (the reference's DDP wrappers also all-reduce the
                # discriminator gradients of the generator phase, which the next zero_grad throws away)
                arena.enable_data_parallel()
            arena_nets.append(net)
            return ArenaAdam(arena, **oparams)
        return optimizer_builder(net.parameters(), otype, oparams)

    for model_name in config["Model"].keys():
        conf = config["Model"][model_name]
        if model_name == "Generator":
            net = _h.Generator(**conf["params"]).to(device)
            model["generator"] = net
            optimizer["generator"] = _opt(net, conf)
            scheduler["generator"] = scheduler_builder(optimizer["generator"], conf["scheduler"].get("type", "StepLR"),
                                                       conf["scheduler"].get("params", {}))
        else:
            net = getattr(_h, model_name)(**conf["params"]).to(device)
            model["discriminator"][model_name] = net
            optimizer["discriminator"][model_name] = _opt(net, conf)
            scheduler["discriminator"][model_name] = scheduler_builder(
                optimizer["discriminator"][model_name], conf["scheduler"].get("type", "StepLR"),
                conf["scheduler"].get("params", {}))
    out_channels = config["Model"]["Generator"]["params"].get("out_channels", 1)
    if out_channels > 1:  # multi-band generator: reference models/__init__.py:64-68
        from kantts.models.pqmf import PQMF

        model["pqmf"] = PQMF(subbands=out_channels, **config.get("pqmf", {})).to(device)
    if distributed:  # networks without a gradient arena (non-Adam optimizers, CPU) fall back to torch DDP
        dev_ids = [rank] if _is_hip(device) else None
        if not any(model["generator"] is n for n in arena_nets):
            model["generator"] = DistributedDataParallel(model["generator"], device_ids=dev_ids, output_device=dev_ids and rank,
                                                         broadcast_buffers=False)
        for name in model["discriminator"].keys():
            if not any(model["discriminator"][name] is n for n in arena_nets):
                model["discriminator"][name] = DistributedDataParallel(
                    model["discriminator"][name], device_ids=dev_ids, output_device=dev_ids and rank,
                    broadcast_buffers=False)
    return model, optimizer, scheduler


def sybert_model_builder(config, device, rank, distributed):
    model, optimizer, scheduler = {}, {}, {}
    conf = config["Model"]["KanTtsTextsyBERT"]
    model["KanTtsTextsyBERT"] = KanTtsTextsyBERT(conf["params"]).to(device)
    optimizer["KanTtsTextsyBERT"] = optimizer_builder(model["KanTtsTextsyBERT"].parameters(),
                                                      conf["optimizer"].get("type", "Adam"),
                                                      conf["optimizer"].get("params", {}))
    scheduler["KanTtsTextsyBERT"] = scheduler_builder(optimizer["KanTtsTextsyBERT"],
                                                      conf["scheduler"].get("type", "StepLR"),
                                                      conf["scheduler"].get("params", {}))
    if distributed:
        model["KanTtsTextsyBERT"] = DistributedDataParallel(model["KanTtsTextsyBERT"], device_ids=[rank],
                                                            output_device=rank)
    return model, optimizer, scheduler


model_dict = {
    "hifigan": hifigan_model_builder,
    "sambert": sambert_model_builder,
    "sybert": sybert_model_builder,
}


def model_builder(config, device="cpu", rank=0, distributed=False):
    builder_func = model_dict[config["model_type"]]
    return builder_func(config, device, rank, distributed)


def __getattr__(name):  # lazy: the HiFi-GAN classes import their own kernels
    if name in ("Generator", "MultiScaleDiscriminator", "MultiPeriodDiscriminator", "MultiSpecDiscriminator",
                "SpecDiscriminator"):
        from kantts.models.hifigan import hifigan as _h

        return getattr(_h, name)
    if name == "PQMF":
        from kantts.models.pqmf import PQMF

        return PQMF
    raise AttributeError(name)
