"""Mirror of the checkpoint functions of yolo/procedures/initialize.py (:12-25 save_model, :57-104 load_checkpoint): the `.tar` files
are the reference's own dictionary, so a run can be resumed by either implementation:

    {'epoch', 'model_state_dict', 'optimizer_state_dict', 'scheduler_state_dict', 'optimizer_name', 'scheduler_name', 'metrics'}

model_state_dict carries the reference's keys and layouts (YoloHead.state_dict); optimizer_state_dict is what torch.optim.SGD / Adam over
`model.parameters()` would save (per-parameter momentum buffers in OIHW), produced from the flat optimizer's single buffer."""
import os
from collections import OrderedDict

import torch


def save_model(model, optimizer, scheduler, metrics, epoch, name, directory="checkpoints"):
    """initialize.py:12-25.  `model` is the YoloHead mirror (or a wrapper with `.module`), `optimizer` a FlatSGD / FlatAdam."""
    os.makedirs(directory, exist_ok=True)
    net = getattr(model, "module", model)
    checkpoint = os.path.join(directory, f"{name}.tar")
    torch.save({"epoch": epoch,
                "model_state_dict": OrderedDict((k, v.cpu()) for k, v in net.state_dict().items()),
                "optimizer_state_dict": _cpu(optimizer.reference_state_dict(net.engine)),
                "scheduler_state_dict": scheduler.state_dict() if scheduler is not None else None,
                "optimizer_name": getattr(optimizer, "name", type(optimizer).__name__),
                "scheduler_name": getattr(scheduler, "name", None),
                "metrics": metrics}, checkpoint)
    return checkpoint


def _cpu(sd):
    return {"state": {i: {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in st.items()} for i, st in sd["state"].items()},
            "param_groups": sd["param_groups"]}


def load_checkpoint(model, optimizer, path, scheduler=None, map_location=None):
    """initialize.py:57-87 for an explicit file: -> (metrics, next epoch).  A `module.` prefix (DataParallel / DDP checkpoints) is
    stripped, as the reference's RuntimeError branch does."""
    if not os.path.exists(path):
        print("checkpoint not found, returning random model")                       # initialize.py:99-104
        return {"mAP": None, "val_loss": None}, 0
    checkpoint = torch.load(path, map_location=map_location or "cpu", weights_only=False)
    net = getattr(model, "module", model)
    net.load_state_dict(checkpoint["model_state_dict"])
    if optimizer is not None and checkpoint.get("optimizer_state_dict") is not None:
        optimizer.load_reference_state_dict(net.engine, checkpoint["optimizer_state_dict"])
        optimizer.name = checkpoint.get("optimizer_name")
    if scheduler is not None:
        try:
            scheduler.load_state_dict(checkpoint["scheduler_state_dict"])
            scheduler.name = checkpoint.get("scheduler_name")
        except Exception:  # noqa: BLE001  (the reference swallows this too, :81-86)
            print("Warning: could not load scheduler")
    return checkpoint["metrics"], checkpoint["epoch"] + 1
