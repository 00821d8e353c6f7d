"""yolo/nets/backbone registry mirror (yolo/nets/backbone/__init__.py:3-6)."""
from ..engine import BLOCKS

backbone_fn = {name: name for name in BLOCKS}
