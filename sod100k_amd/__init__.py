"""sod100k_amd -- MI355X-native CSNet engine (hand-written HIP for gfx950 behind a C ABI).

Layout:
  csrc/            HIP kernels + the C-ABI library (libcsnet_hip.so, see include/csnet_hip.h)
  _native.py       ctypes binding + build recipe (no fallback: raises when the library is missing)
  engine.py        plan construction from a CSNet module tree, workspace + parameter arena
  model/           drop-in mirror of the reference's ``model`` package (csnet.py, conv2d.py, utils/)
  configs/         yacs-free config loader accepting the reference's YAML files
  tools/           test.py / train.py counterparts of the reference's caller scripts
  data/            the shipped checkpoints re-encoded (raw blob + JSON manifest)
"""
__version__ = "0.1.0"
