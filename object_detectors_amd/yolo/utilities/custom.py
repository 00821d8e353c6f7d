"""Mirror of the class re-weighting helper of yolo/utilities/custom.py: `IDFTransformer` (:163-262).

The reference builds a table of per-class statistics (document / instance frequencies and several idf variants) from the annotation file
and caches it as `<owd>/<dset>_files/idf.csv` (shipped in the repository for COCO and LVIS); YOLOForw reads class weights / logit
multipliers from it (yolo_forw.py:35-67) and calls `forward(targets)` for the per-batch idf row when `tfidf_batch` is set (:87-91).
Only the cached-table path is provided here (building it needs pycocotools / lvis, which is data-pipeline work outside the hot path)."""
import csv
import os

import torch
import torch.nn as nn


class IDFTransformer(nn.Module):
    def __init__(self, annfile=None, dset_name="coco", device="cuda", reduce="sum", reduce_mini_batch=True, csv_path=None, num_classes=None):
        super().__init__()
        self.device = device
        self.idf_weights = {}
        if csv_path is None:
            owd = os.getenv("owd")
            if owd is not None:
                csv_path = os.path.join(owd, dset_name + "_files", "idf.csv")
        if csv_path is not None and os.path.exists(csv_path):
            with open(csv_path, newline="") as f:
                rows = list(csv.DictReader(f))
            for k in rows[0].keys():
                try:
                    vals = [float(r[k]) for r in rows]
                except ValueError:
                    continue                                   # string columns are skipped, as the reference does (custom.py:251-252)
                self.idf_weights[k] = torch.tensor(vals, dtype=torch.float32, device=device)
            self.num_classes = len(rows)
        elif num_classes is not None:
            self.num_classes = num_classes                     # per-batch idf only
        else:
            raise FileNotFoundError("IDFTransformer: no cached idf.csv (set $owd or pass csv_path); building the table from annotations needs "
                                    "pycocotools / lvis and is not part of this path")

    def forward(self, targets):
        """custom.py:257-262: smoothed idf of the classes present in this mini-batch."""
        t = torch.stack([torch.bincount(x["category_id"], minlength=self.num_classes) for x in targets])
        t[t > 0] = 1
        t = t.sum(axis=0)
        return torch.log((len(targets) + 1) / (t + 1)) + 1
