"""importlib-by-name factories with the reference's discovery rules (utils/initialization.py:48-117,
training/trainer.py:155-158): the drop-in surface of this repository.

    create_alg(algorithm="DSAC_V2_HIP", **kw)          -> module dsac_v2_hip, class DSAC_V2_HIP
    create_buffer(buffer_name="hip_replay_buffer", **kw) -> module training.hip_replay_buffer,
                                                            class HipReplayBuffer
    create_trainer(alg, sampler, buffer, evaluator, **kw) -> training.hip_trainer.HipOffSerialTrainer
"""
import importlib
import os
import sys

_PKG = os.path.dirname(os.path.abspath(__file__))
if _PKG not in sys.path:
    sys.path.insert(0, _PKG)


def install():
    """One call for a reference checkout: makes `training.hip_replay_buffer` (and the other training/hip_*.py modules)
    importable INSIDE the reference's own `training` package, which is where `utils/initialization.py:93-110`
    (`importlib.import_module("training." + buffer_name)`) looks for a buffer -- without copying files into the
    reference. Works whichever `training` package was imported first: the package's search path ends up holding
    both directories. `dsac_v2_hip` / `dsac_v1_hip` need nothing: they are top-level modules of this directory."""
    import training

    ours = os.path.join(_PKG, "training")
    dirs = [ours] + [os.path.join(p, "training") for p in sys.path if p and os.path.isdir(os.path.join(p, "training"))]
    for d in dirs:
        if d not in list(training.__path__):
            training.__path__.append(d)
    importlib.invalidate_caches()


def camel(name: str) -> str:
    return "".join(part[:1].upper() + part[1:] for part in name.split("_"))


def create_alg(**kwargs):
    name = kwargs["algorithm"]
    module = importlib.import_module(name.lower())
    if not hasattr(module, name):
        raise NotImplementedError("algorithm %s is not defined in module %s" % (name, name.lower()))
    return getattr(module, name)(**kwargs)


def create_buffer(**kwargs):
    file_name = kwargs["buffer_name"].lower()
    module = importlib.import_module("training." + file_name)
    cls = camel(file_name)
    if not hasattr(module, cls):
        raise NotImplementedError("buffer %s is not defined in training.%s" % (cls, file_name))
    return getattr(module, cls)(**kwargs)


def create_env(**kwargs):
    """reference utils/initialization.py:9-45: module `<env_id>_data` exposing `env_creator(**kwargs)`
    (or the CamelCase class); the environment itself is outside this repository's scope."""
    name = kwargs["env_id"]
    module = importlib.import_module(name + "_data")
    if hasattr(module, "env_creator"):
        return module.env_creator(**kwargs)
    if hasattr(module, camel(name)):
        return getattr(module, camel(name))(**kwargs)
    raise NotImplementedError("environment %s is not properly defined" % name)


def create_sampler(**kwargs):
    from training.hip_sampler import HipOffSampler

    return HipOffSampler(**kwargs)


def create_evaluator(**kwargs):
    from training.hip_trainer import HipEvaluator

    return HipEvaluator(**kwargs)


def save_tb_to_csv(path):
    """reference utils/tensorboard_setup.py:121-139 for a folder written by HipOffSerialTrainer"""
    from training.hip_trainer import save_tb_to_csv as f

    return f(path)


def create_trainer(alg, sampler, buffer, evaluator, **kwargs):
    from training.hip_trainer import HipOffSerialTrainer

    return HipOffSerialTrainer(alg, sampler, buffer, evaluator, **kwargs)
