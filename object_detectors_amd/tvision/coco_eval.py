"""Mirror of the detection-to-COCO formatting of torchvision_models/detection/coco_eval.py (:83-105 prepare_for_coco_detection,
:169-171 convert_to_xywh): the wire format handed to pycocotools / lvis."""
import torch

from .._lib import check, lib, ptr, stream_ptr


def convert_to_xywh(boxes):
    """coco_eval.py:169-171."""
    k = int(boxes.shape[0])
    b = boxes.float().contiguous()
    out = torch.empty((k, 4), dtype=torch.float32, device=b.device)
    check(lib().mi355det_coco_rows(ptr(b), 4, None, None, 0, k, 1.0, 1.0, 1.0, 0, 2, ptr(out), None, None, stream_ptr()), "coco_rows")
    return out


def prepare_for_coco_detection(predictions):
    """coco_eval.py:83-105: {image_id: {'boxes' [k,4] xyxy, 'scores' [k], 'labels' [k]}} -> list of result dicts."""
    coco_results = []
    for original_id, prediction in predictions.items():
        if len(prediction) == 0:
            continue
        boxes = convert_to_xywh(prediction["boxes"]).tolist()
        scores = prediction["scores"].tolist()
        labels = prediction["labels"].tolist()
        coco_results.extend([{"image_id": original_id, "category_id": labels[k], "bbox": box, "score": scores[k]} for k, box in enumerate(boxes)])
    return coco_results
