"""Reader for the reference's training configuration files (configs/config.yaml, config-4.yaml, config-256.yaml).

ddp-train.py:592-595 loads the YAML into a yacs CfgNode and reads `train.init_learning_rate` (:78), `train.lr_scheduler_iter`
and `train.gamma` (:82), `train.clip` (:441), `train.batchsize` (:321), `train.losses` (:404-430), `train.nids`,
`train.maxiter`, `train.num_epochs`.  This module reads the same keys (plain PyYAML -- yacs is not needed for that) into
the keyword arguments of `trainloop.Trainer`, so that a maintainer can point the loop of this build at the reference's own
file: `Trainer.from_config(model, load_train_config("configs/config-4.yaml"))`, `bench.py --mode train --config <file>`.
Keys that belong to parts of ava-256 outside this build (dataset paths, data loader, progress / tensorboard) are kept
under "other" and otherwise ignored.  `opts` = the reference's `--opts key value ...` overrides (`merge_from_list`, :596).
"""
import yaml

# what the reference's own file says (configs/config.yaml:9-21); used for keys a config leaves out
DEFAULTS = {"init_learning_rate": 2.0e-4, "lr_scheduler_iter": 10_000, "gamma": 1.4, "clip": 1.0, "batchsize": 4,
            "losses": {"irgbl1": 1.0, "vertl1": 0.1, "kldiv": 1.0e-3, "primvolsum": 0.01}}


def _set_path(d, dotted, value):
    keys = dotted.split(".")
    for k in keys[:-1]:
        d = d.setdefault(k, {})
    d[keys[-1]] = value


def load_train_config(path, opts=()):
    """Returns {"trainer": kwargs of Trainer, "batchsize": frames per GPU, "nids", "maxiter", "num_epochs", "other": {...}}."""
    with open(path, "r") as f:
        doc = yaml.safe_load(f) or {}
    opts = list(opts)
    if len(opts) % 2:
        raise ValueError("--opts takes key value pairs")
    for k, v in zip(opts[0::2], opts[1::2]):
        _set_path(doc, k, yaml.safe_load(v) if isinstance(v, str) else v)
    tr = doc.get("train")
    if not isinstance(tr, dict):
        raise KeyError("%s has no `train` section" % path)
    get = lambda k: tr.get(k, DEFAULTS[k])
    losses = get("losses")
    if not isinstance(losses, dict) or not losses:
        raise ValueError("train.losses must map loss names to weights")
    known = {"irgbl1", "vertl1", "kldiv", "primvolsum"}
    if set(losses) - known:
        raise NotImplementedError("loss term(s) %s are not on this build's path (it has %s)" % (
            sorted(set(losses) - known), sorted(known)))
    trainer = {"lr": float(get("init_learning_rate")), "lr_scheduler_iter": int(get("lr_scheduler_iter")),
               "gamma": float(get("gamma")), "clip": float(get("clip")),
               "loss_weights": {k: float(v) for k, v in losses.items()}}
    if not (trainer["lr"] > 0 and trainer["lr_scheduler_iter"] > 0 and trainer["gamma"] > 0 and trainer["clip"] > 0):
        raise ValueError("train.{init_learning_rate, lr_scheduler_iter, gamma, clip} must be positive")
    used = {"init_learning_rate", "lr_scheduler_iter", "gamma", "clip", "batchsize", "losses", "nids", "maxiter", "num_epochs"}
    other = {k: v for k, v in tr.items() if k not in used}
    other.update({k: v for k, v in doc.items() if k != "train"})
    return {"trainer": trainer, "batchsize": int(get("batchsize")), "nids": tr.get("nids"), "maxiter": tr.get("maxiter"),
            "num_epochs": tr.get("num_epochs"), "other": other}
