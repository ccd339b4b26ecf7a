"""Aliases for the stale flat-layout spellings the reference's own harnesses still import
(SURVEY.md 0 and 8(b)): `from utils import CTCLoss` (benchmarks/ctc_benchmark.py:13),
`from utils import ASGLoss` (benchmarks/asg_benchmark.py:13), `import transducer`
(benchmarks/transducer_benchmark.py:13, tests/transducer_test.py:17-19),
`utils.pack_replabels` (tests/utils_test.py:19), `from criterions import ctc, asg, transducer`
(utils.py:19), `utils.load_criterion` (utils.py:245-273; callers train.py:201, test.py:75), and `models.load_criterion` /
`models.load_from_checkpoint` (test.py:77-86; `models.load_model` builds acoustic models and is the caller's).  `install()` registers them in `sys.modules`; nothing is copied or patched on disk.
"""
import sys
import types


def install(gtn_alias=False):
    from . import criterions, graph
    from .criterions import asg, ctc, stc, transducer

    sys.modules["criterions"] = criterions
    for name, mod in (("ctc", ctc), ("asg", asg), ("stc", stc), ("transducer", transducer)):
        sys.modules["criterions." + name] = mod
    sys.modules["transducer"] = transducer
    utils = sys.modules.get("utils")
    if utils is None:
        utils = types.ModuleType("utils")
        utils.__doc__ = "criterion names of the reference's former flat layout (gtn_applications_amd.compat)"
        sys.modules["utils"] = utils
    from . import load_criterion, load_from_checkpoint

    models = sys.modules.get("models")
    if models is None:
        models = types.ModuleType("models")
        models.__doc__ = "criterion-side names test.py expects under `models` (gtn_applications_amd.compat)"
        sys.modules["models"] = models
    for name, obj in dict(load_criterion=load_criterion, load_from_checkpoint=load_from_checkpoint).items():
        if not hasattr(models, name):
            setattr(models, name, obj)
    for name, obj in dict(
        load_criterion=load_criterion, load_from_checkpoint=load_from_checkpoint, CTCLoss=ctc.CTCLoss, CTCLossFunction=ctc.CTCLossFunction, ASGLoss=asg.ASGLoss,
        ASGLossFunction=asg.ASGLossFunction, pack_replabels=asg.pack_replabels,
        unpack_replabels=asg.unpack_replabels, STCLoss=stc.STCLoss,
    ).items():
        if not hasattr(utils, name):
            setattr(utils, name, obj)
    if gtn_alias and "gtn" not in sys.modules:
        # graph construction / text I/O subset of the gtn API (Graph, add_node, add_arc, compose,
        # remove, project_*, loadtxt, ...).  Scoring functions are NOT graph-level here: they live
        # in the criteria (device engine).
        sys.modules["gtn"] = graph
    return utils
