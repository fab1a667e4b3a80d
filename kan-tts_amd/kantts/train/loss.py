"""Training criteria (reference kantts/train/loss.py).  The masked-L1 reductions of the SAM-BERT
step run in one kernel each (kantts_masked_l1: loss scalar + d loss/d pred in the same pass)."""
import torch
import torch.nn.functional as F

from kantts._hip import ops


class MelReconLoss(torch.nn.Module):
    """Masked mean |target - output| for decoder and postnet mels (reference :7-37)."""

    def __init__(self, loss_type="mae"):
        super(MelReconLoss, self).__init__()
        self.loss_type = loss_type
        if loss_type != "mae":
            raise NotImplementedError("only loss_type='mae' is used by the shipped configs")

    def forward(self, output_lengths, mel_targets, dec_outputs, postnet_outputs=None):
        lens = output_lengths.to(torch.int64)
        mel_loss_ = ops.masked_l1(dec_outputs, mel_targets, lens)
        mel_loss = ops.masked_l1(postnet_outputs, mel_targets, lens) if postnet_outputs is not None else 0.0
        return mel_loss_, mel_loss


class ProsodyReconLoss(torch.nn.Module):
    """Masked L1 on log-duration, pitch, energy (reference :40-85)."""

    def __init__(self, loss_type="mae"):
        super(ProsodyReconLoss, self).__init__()
        self.loss_type = loss_type
        if loss_type != "mae":
            raise NotImplementedError("only loss_type='mae' is used by the shipped configs")

    def forward(self, input_lengths, duration_targets, pitch_targets, energy_targets, log_duration_predictions,
                pitch_predictions, energy_predictions):
        lens = input_lengths.to(torch.int64)
        dur_loss = ops.masked_l1(log_duration_predictions, torch.log(duration_targets.float() + 1), lens)
        pitch_loss = ops.masked_l1(pitch_predictions, pitch_targets, lens)
        energy_loss = ops.masked_l1(energy_predictions, energy_targets, lens)
        return dur_loss, pitch_loss, energy_loss


loss_dict = {
    "MelReconLoss": MelReconLoss,
    "ProsodyReconLoss": ProsodyReconLoss,
}


def criterion_builder(config, device="cpu"):
    """Same contract as reference :528-544: dict of enabled criteria + ``.weights`` entries."""
    criterion = {}
    for key, value in config["Loss"].items():
        if key in loss_dict:
            if value["enable"]:
                criterion[key] = loss_dict[key](**value.get("params", {})).to(device)
                setattr(criterion[key], "weights", value.get("weights", 1.0))
        else:
            raise NotImplementedError("{} is not implemented".format(key))
    return criterion
