"""Import shim for the READ-ONLY reference checkout (authoring container only).

The reference hot path (src.models.net.*, src.models.score.*, src.common.*) imports
cleanly once two things it does not need at run time are stubbed:
  * ``tree`` (dm-tree): residue_constants.py:23,739 uses ``tree.map_structure`` once.
  * ``src.utils`` package __init__ (imports hydra/lightning): pre-registered as a bare
    package so ``src.utils.tensor_utils`` still resolves.
Nothing here travels to the GPU box: only ``make_golden.py`` (fixture generator) and
``oracle/validate_against_reference.py`` use it, both in the authoring container.
"""
import os
import sys
import types

REF_ROOT = os.environ.get("STR2STR_REFERENCE", "/root/reference")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REF_ROOT, "src", "models"))


def install():
    if not reference_available():
        raise RuntimeError(f"reference checkout not found at {REF_ROOT}")
    if "tree" not in sys.modules:
        tree = types.ModuleType("tree")

        def map_structure(fn, *structs):
            s0 = structs[0]
            if isinstance(s0, dict):
                return {k: map_structure(fn, *[s[k] for s in structs]) for k in s0}
            if isinstance(s0, (list, tuple)):
                return type(s0)(map_structure(fn, *xs) for xs in zip(*structs))
            return fn(*structs)

        tree.map_structure = map_structure
        sys.modules["tree"] = tree
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    # Make sure ``src`` resolves to the reference (no package of that name lives in this repo).
    import importlib

    if "src" in sys.modules and not getattr(sys.modules["src"], "__file__", "").startswith(REF_ROOT):
        raise RuntimeError("a different 'src' package is already imported")
    src = importlib.import_module("src")
    if "src.utils" not in sys.modules:
        utils = types.ModuleType("src.utils")
        utils.__path__ = [os.path.join(REF_ROOT, "src", "utils")]
        sys.modules["src.utils"] = utils
        setattr(src, "utils", utils)
    return src
