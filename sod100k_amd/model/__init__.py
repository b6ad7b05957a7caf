"""Mirror of the reference's ``model`` package (CSNet/model/): import as ``sod100k_amd.model.csnet`` or put
``sod100k_amd/`` on ``sys.path`` next to the repo root so that ``importlib.import_module("model.csnet")``
(test.py:37, train.py:70) resolves here."""
