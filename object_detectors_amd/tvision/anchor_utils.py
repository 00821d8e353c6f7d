"""Mirror of torchvision_models/tvision/anchor_utils.py:AnchorGenerator (anchor_utils.py:10-159)."""
import torch
import torch.nn as nn

from .. import ops


class AnchorGenerator(nn.Module):
    def __init__(self, sizes=((128, 256, 512),), aspect_ratios=((0.5, 1.0, 2.0),)):
        super().__init__()
        if not isinstance(sizes[0], (list, tuple)):
            sizes = tuple((s,) for s in sizes)
        if not isinstance(aspect_ratios[0], (list, tuple)):
            aspect_ratios = (aspect_ratios,) * len(sizes)
        assert len(sizes) == len(aspect_ratios)
        self.sizes, self.aspect_ratios = sizes, aspect_ratios
        self.cell_anchors = None

    def generate_anchors(self, scales, aspect_ratios, dtype=torch.float32, device=torch.device("cpu")):
        # anchor_utils.py:60-71 — tiny host-side table (A rows), float32 like the reference
        scales = torch.as_tensor(scales, dtype=dtype)
        aspect_ratios = torch.as_tensor(aspect_ratios, dtype=dtype)
        h_ratios = torch.sqrt(aspect_ratios)
        w_ratios = 1 / h_ratios
        ws = (w_ratios[:, None] * scales[None, :]).view(-1)
        hs = (h_ratios[:, None] * scales[None, :]).view(-1)
        return (torch.stack([-ws, -hs, ws, hs], dim=1) / 2).round().to(device)

    def set_cell_anchors(self, dtype, device):
        if self.cell_anchors is not None and self.cell_anchors[0].device == device:
            return
        self.cell_anchors = [self.generate_anchors(s, a, dtype, device) for s, a in zip(self.sizes, self.aspect_ratios)]

    def num_anchors_per_location(self):
        return [len(s) * len(a) for s, a in zip(self.sizes, self.aspect_ratios)]

    def grid_anchors(self, grid_sizes, strides):
        if not (len(grid_sizes) == len(strides) == len(self.cell_anchors)):
            raise ValueError("Anchors should be Tuple[Tuple[int]] because each feature map could potentially have "
                             "different sizes and aspect ratios.")
        return [ops.anchor_grid(base, int(gh), int(gw), int(sh), int(sw))
                for (gh, gw), (sh, sw), base in zip(grid_sizes, strides, self.cell_anchors)]

    def forward(self, image_list, feature_maps):
        grid_sizes = [fm.shape[-2:] for fm in feature_maps]
        image_size = image_list.tensors.shape[-2:]
        device = feature_maps[0].device
        strides = [[image_size[0] // g[0], image_size[1] // g[1]] for g in grid_sizes]
        self.set_cell_anchors(torch.float32, device)
        per_level = self.grid_anchors(grid_sizes, strides)
        allc = torch.cat(per_level)
        return [allc for _ in range(len(image_list.image_sizes))]
