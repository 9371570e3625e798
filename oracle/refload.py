"""Load the reference for golden-vector generation / oracle validation.

TEST INFRASTRUCTURE ONLY (build container; /root/reference never travels).

* `load_ref_bpmodel()`  -> the compiled reference kernel module built by
  oracle/build_ref.py (binary only, build container only: `.gpurunignore`
  lists oracle/_ref/, so the reference never goes to the GPU box in any form;
  the bench's CPU baseline there is the oracle port, `kind: "port"`).
* `load_ref_cn_model()` -> the reference's Python host class module
  (`remixt/cn_model.py`), imported from /root/reference *in place*.  Only
  possible in the build container.

A synthetic `remixt` package object is registered whose __path__ spans
oracle/_ref/remixt (the binary) and /root/reference/remixt (python sources), so
the reference's package __init__ (versioneer) is not needed and nothing is
copied.  `statsmodels.tools.numdiff` is imported by cn_model.py:9 but only used
in a failure-diagnostic branch (cn_model.py:513-518); it is absent from this
image, so an empty placeholder module is registered for the import to succeed.
"""
import importlib
import os
import sys
import types

# importing the reference's .py files in place must not leave __pycache__ files in /root/reference
# (the tree is not ours to write to)
sys.dont_write_bytecode = True

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("REMIXT_REFERENCE", "/root/reference")
REF_BIN = os.path.join(HERE, "_ref", "remixt")


def _placeholder_if_absent(names):
    """Register an empty module for every name that cannot be imported (never shadows a real package)."""
    for name in names:
        if name in sys.modules:
            continue
        try:
            importlib.import_module(name)
        except Exception:
            sys.modules[name] = types.ModuleType(name)


def _ensure_pkg(with_sources):
    pkg = sys.modules.get("remixt")
    if pkg is None or not getattr(pkg, "_oracle_synthetic", False):
        pkg = types.ModuleType("remixt")
        pkg._oracle_synthetic = True
        pkg.__path__ = []
        sys.modules["remixt"] = pkg
    paths = [REF_BIN]
    if with_sources:
        paths.append(os.path.join(REF, "remixt"))
    for p in paths:
        if p not in pkg.__path__:
            pkg.__path__.append(p)
    return pkg


def have_ref_binary():
    return os.path.isdir(REF_BIN) and any(
        f.startswith("bpmodel") and f.endswith(".so") for f in os.listdir(REF_BIN))


def have_ref_sources():
    return os.path.exists(os.path.join(REF, "remixt", "cn_model.py"))


def load_ref_bpmodel():
    if not have_ref_binary():
        raise ImportError("oracle/_ref not built (run oracle/build_ref.py)")
    _ensure_pkg(with_sources=False)
    return importlib.import_module("remixt.bpmodel")


def load_ref_cn_model():
    if not have_ref_sources():
        raise ImportError("reference sources not present")
    load_ref_bpmodel()
    _ensure_pkg(with_sources=True)
    _placeholder_if_absent(("statsmodels", "statsmodels.tools", "statsmodels.tools.numdiff"))
    return importlib.import_module("remixt.cn_model")


def load_ref_analysis():
    """The reference's read-depth initialisation / result-table modules, imported in place:
    (likelihood, analysis.experiment, analysis.readdepth, analysis.pipeline).  `pypeliner` is imported
    by remixt/utils.py:10 and unused on this path; it is absent from this image, so an empty placeholder
    module is registered for the import to succeed."""
    load_ref_cn_model()
    _placeholder_if_absent(("pypeliner", "pypeliner.commandline"))
    return tuple(importlib.import_module(m) for m in
                 ("remixt.likelihood", "remixt.analysis.experiment", "remixt.analysis.readdepth", "remixt.analysis.pipeline"))


def load_ref_simulations():
    """The reference's simulation module (`remixt/simulations/experiment.py`), imported in place, for the
    genome-mixture and read-count samplers (:965-1399).  `remixt/simulations/balanced.py` imports
    `networkx` and `blossomv.blossomv` at module level, only used by `collapsed_balanced_breakpoints`;
    `blossomv` is absent from this image, so an empty placeholder module is registered for the import to
    succeed (a package that is importable is never shadowed).  The rearrangement-history sampler in the same file calls `scipy.misc.logsumexp`
    (:698), which no longer exists: that part is not usable here and no vectors come from it."""
    load_ref_analysis()
    _placeholder_if_absent(("networkx", "blossomv", "blossomv.blossomv"))
    return importlib.import_module("remixt.simulations.experiment")


def load_ref_evaluation():
    """The reference's accuracy statistics (`remixt/simulations/pipeline.py:343-647`, `evaluate_results`
    and the two functions under it; `remixt/segalg.py:260-336` `reindex_segments`), imported in place.
    The module also imports the read simulator, whose `remixt/seqdataio.py` needs the reference's compiled
    `remixt.bamreader` (htslib sources, not buildable here) -- unused by the evaluation functions, so an
    empty placeholder module is registered for that import."""
    load_ref_simulations()
    _placeholder_if_absent(("remixt.bamreader",))
    return importlib.import_module("remixt.simulations.pipeline")
