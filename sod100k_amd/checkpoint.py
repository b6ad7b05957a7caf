"""Checkpoint IO: the reference's ``torch.save({'epoch','arch','state_dict','optimizer'})`` files
(train.py:172-181) and this repo's neutral re-encoding of the shipped checkpoints
(``data/<name>.json`` manifest + ``<name>.bin`` raw little-endian blob, written by oracle/make_goldens.py)."""
import json
import os
from typing import Dict

import numpy as np
import torch


def load_manifest_state_dict(manifest_path: str) -> Dict[str, torch.Tensor]:
    with open(manifest_path) as f:
        man = json.load(f)
    blob = np.fromfile(os.path.join(os.path.dirname(os.path.abspath(manifest_path)), man["blob"]), dtype=np.uint8)
    sd = {}
    for e in man["tensors"]:
        n = int(np.prod(e["shape"])) if len(e["shape"]) else 1
        arr = np.frombuffer(blob, dtype=np.dtype(e["dtype"]), count=n, offset=e["offset"]).reshape(e["shape"])
        sd[e["name"]] = torch.from_numpy(arr.copy())
    return sd


def load_checkpoint(path: str, map_location="cpu") -> dict:
    """Either format -> ``{'epoch', 'arch', 'state_dict', 'optimizer'}``."""
    if str(path).endswith(".json"):
        with open(path) as f:
            man = json.load(f)
        return dict(epoch=man.get("epoch", 0), arch=man.get("arch", "csnet"),
                    state_dict=load_manifest_state_dict(path), optimizer=None)
    return torch.load(path, map_location=map_location)


def save_checkpoint(state: dict, filename: str) -> None:
    torch.save(state, filename)     # train.py:296-297
