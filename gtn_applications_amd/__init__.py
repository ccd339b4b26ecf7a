"""MI355X-native differentiable-WFST loss engine: drop-in criteria for the hot path of
facebookresearch/gtn_applications (criterions/{ctc,asg,stc,transducer}.py), backed by hand-written
gfx950 HIP kernels behind the C ABI of include/wfl.h (libwfl.so).  See DESIGN.md."""
from . import _native  # noqa: F401  (fails loudly if libwfl.so is missing)
from . import graph  # noqa: F401

__all__ = ["graph", "criterions", "engine", "load_criterion", "load_from_checkpoint"]


def load_criterion(criterion_type, preprocessor, config):
    """The reference's criterion factory (utils.py:245-273): returns (criterion, output_size).
    `preprocessor` needs `num_tokens`, and for the transducer `tokens` and `graphemes_to_index`."""
    from . import graph as G
    from .criterions import asg, ctc, transducer

    num_tokens = preprocessor.num_tokens
    if criterion_type == "asg":
        num_replabels = config.get("num_replabels", 0)
        use_garbage = config.get("use_garbage", True)
        return (asg.ASG(num_tokens, num_replabels, use_garbage), num_tokens + num_replabels + int(use_garbage))
    elif criterion_type == "ctc":
        use_pt = config.get("use_pt", True)  # use pytorch implementation
        return ctc.CTC(num_tokens, use_pt), num_tokens + 1  # account for blank
    elif criterion_type == "transducer":
        blank = config.get("blank", "none")
        transitions = config.get("transitions", None)
        if transitions is not None:
            # utils.py:261 reads this file with gtn.load.  G.load accepts gtn's text format (pinned by the
            # reference's tests/trans_backoff_test.txt) and gtn's binary layout as restated in include/wfl.h
            # (UNPINNED: gtn is not vendored); a file that is consistently neither raises instead of being mis-read.
            transitions = G.load(transitions)
        criterion = transducer.Transducer(
            preprocessor.tokens,
            preprocessor.graphemes_to_index,
            ngram=config.get("ngram", 0),
            transitions=transitions,
            blank=blank,
            allow_repeats=config.get("allow_repeats", True),
            reduction="mean",
        )
        return criterion, num_tokens + int(blank != "none")
    else:
        raise ValueError(f"Unknown model type {criterion_type}")


def load_from_checkpoint(model, criterion, checkpoint_path, load_last=False):
    """utils.py:276-283: restores `model.checkpoint[.best]` / `criterion.checkpoint[.best]`; the criteria's
    parameter names (`transitions`, `transition_params`) are the reference's, so its checkpoints load."""
    import os

    import torch

    suffix = "" if load_last else ".best"
    model.load_state_dict(torch.load(os.path.join(checkpoint_path, "model.checkpoint" + suffix)))
    criterion.load_state_dict(torch.load(os.path.join(checkpoint_path, "criterion.checkpoint" + suffix)))
