"""yacs-free configuration tree accepting the reference's YAML files unchanged.

The reference builds a ``yacs.config.CfgNode`` in ``configs/defaults.py`` (CSNet/configs/defaults.py:1-89;
the training copy adds PRUNE / AUTO / FINETUNE, CSNet_training/configs/defaults.py:91-120) and merges a YAML
file over it with ``cfg.merge_from_file`` (test.py:23-25).  yacs is not a dependency here: ``CfgNode`` below
keeps the same surface (attribute access, ``merge_from_file``, unknown keys raise ``KeyError`` like yacs).
"""
import ast
import copy

import yaml


def _decode(v):
    """yacs' _decode_cfg_value: strings that are Python literals become the literal (PyYAML reads `1e-4` as str)."""
    if not isinstance(v, str):
        return v
    try:
        return ast.literal_eval(v)
    except (ValueError, SyntaxError):
        return v


class CfgNode(dict):
    def __init__(self, init=None):
        super().__init__()
        for k, v in (init or {}).items():
            self[k] = CfgNode(v) if isinstance(v, dict) else v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def clone(self):
        return copy.deepcopy(self)

    def _merge(self, other, path=""):
        for k, v in other.items():
            if k not in self:
                raise KeyError(f"Non-existent config key: {path + k}")
            if isinstance(self[k], CfgNode):
                if not isinstance(v, dict):
                    raise ValueError(f"{path + k} must be a mapping")
                self[k]._merge(v, path + k + ".")
            else:
                self[k] = _decode(v)

    def merge_from_file(self, filename):
        with open(filename) as f:
            self._merge(yaml.safe_load(f) or {})

    def merge_from_list(self, items):
        for key, val in zip(items[0::2], items[1::2]):
            node = self
            parts = key.split(".")
            for p in parts[:-1]:
                node = node[p]
            if parts[-1] not in node:
                raise KeyError(f"Non-existent config key: {key}")
            node[parts[-1]] = _decode(val)


def defaults() -> CfgNode:
    """The union of the reference's two default trees (same keys, same default values)."""
    ft_solver = dict(METHOD='Adam', MAX_EPOCHS=20, LR=1e-7, MOMENTUM=0.95, WEIGHT_DECAY=5e-3, ADJUST_STEP=False,
                     STEPS=[50, 100], WARMUPLR=0, STEPSIZE=20, GAMMA=0.5, LR_SCHEDULER='step')
    return CfgNode(dict(
        TASK="", GPU=0, PRINT_FREQ=10,
        MODEL=dict(ARCH='csnet', BASIC_SPLIT=[1]),
        LOSS=dict(MLOSS=4),
        DATA=dict(DIR='', BATCH_SIZE=32, WORKERS=4, SAVEDIR='results/', PRETRAIN='', RESUME='', IMAGE_H=224,
                  IMAGE_W=224, AUG=False),
        VAL=dict(DIR='', PRINT_FREQ=20),
        TEST=dict(DATASET_PATH='', BEGIN=200, INTERVAL=5, DATASETS=['ECSSD'], CHECKPOINT='', ENABLE=True, IMAGE_H=0,
                  IMAGE_W=0, TESTALL=False, MODEL_CONFIG=''),
        SOLVER=dict(METHOD='Adam', MAX_EPOCHS=100, LR=1e-4, MOMENTUM=0.95, WEIGHT_DECAY=5e-3, ADJUST_STEP=False,
                    STEPS=[200, 250], WARMUPLR=0, STEPSIZE=20, GAMMA=0.5, LR_SCHEDULER='step',
                    FINETUNE=dict(METHOD='Adam', LR=1e-4, MOMENTUM=0.95, WEIGHT_DECAY=5e-3, GAMMA=0.5,
                                  ADJUST_STEP=False, STEPS=[5, 10], LR_SCHEDULER='step')),
        PRUNE=dict(BNS=False, SHOW=True),
        AUTO=dict(ENABLE=False, PREDEFINE='', FINETUNE=300, FLOPS=dict(ENABLE=False, WEIGHT=0.0, EXPAND=-1.0),
                  EXPAND=1.0, LOAD_WEIGHT="NO"),
        FINETUNE=dict(ENABLE=False, THRES=1e-40, SOLVER=ft_solver),
    ))


cfg = defaults()
