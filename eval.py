"""Entry point with the reference's CLI:  python eval.py task_name=inference [ckpt_path=…] [target_dir=null]
(reference: src/eval.py:102-173).  Composes configs/eval.yaml (hydra if installed, otherwise the built-in
composer), instantiates datamodule / model / trainer from their `_target_`s, loads the checkpoint with the
reference's key contract and runs ``trainer.predict``.  Metrics evaluation (src/eval.py:47-99) is out of
scope and skipped with a log line."""
import logging
import os
import sys

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("PROJECT_ROOT", ROOT)

from str2str_amd.utils import config as C  # noqa: E402

log = logging.getLogger("str2str_amd.eval")


def load_dotenv(path):
    if os.path.exists(path):
        for line in open(path):
            line = line.strip()
            if line and not line.startswith("#") and "=" in line:
                k, _, v = line.partition("=")
                os.environ.setdefault(k.strip(), v.strip().strip('"').strip("'"))


def load_model_checkpoint(model, ckpt_path):
    """.pth -> net weights only ('net.' prefix stripped, strict); .ckpt -> left to the trainer
    (reference src/utils/checkpoint_utils.py:3-27)."""
    if ckpt_path is None:
        return model, None
    if ckpt_path.endswith(".pth"):
        params = torch.load(ckpt_path, map_location=torch.device("cpu"))["state_dict"]
        model.net.load_state_dict({k.replace("net.", ""): v for k, v in params.items()})
        return model, None
    if ckpt_path.endswith(".ckpt"):
        return model, ckpt_path
    raise ValueError(f"ckpt_path {ckpt_path} is not a valid checkpoint file.")


def evaluate(cfg):
    pred_dir = cfg.get("pred_dir")
    if pred_dir and os.path.isdir(pred_dir):
        log.info("pred_dir given: metric evaluation is outside this build's scope (use the reference's eval).")
        return pred_dir
    log.info(f"Instantiating datamodule <{cfg.data['_target_']}>")
    datamodule = C.instantiate(cfg.data)
    log.info(f"Instantiating model <{cfg.model['_target_']}>")
    model = C.instantiate(cfg.model)
    log.info(f"Instantiating trainer <{cfg.trainer['_target_']}>")
    trainer = C.instantiate(cfg.trainer)
    if cfg.get("ckpt_path"):
        model, ckpt_path = load_model_checkpoint(model, cfg.ckpt_path)
    else:
        from str2str_amd.synth import synth_state_dict

        log.warning("ckpt_path is null: using seeded synthetic weights (smoke run, not a trained model)")
        man = [(k, tuple(v.shape)) for k, v in model.net.state_dict().items()]
        model.net.load_state_dict(synth_state_dict(man, seed=0, sigma_final=0.002))
        ckpt_path = None
    datamodule.setup(stage="predict")
    dataloaders = datamodule.test_dataloader()
    if cfg.get("dry_run"):
        log.info(f"dry_run: {len(dataloaders)} target(s) featurised, model + checkpoint ready; not sampling.")
        return None
    log.info("Starting predictions.")
    pred_dir = trainer.predict(model=model, dataloaders=dataloaders, ckpt_path=ckpt_path)[-1]
    log.info(f"Samples written under {pred_dir} (metric evaluation is out of scope of this build).")
    return pred_dir


def main(argv=None):
    logging.basicConfig(level=logging.INFO, format="[%(asctime)s][%(name)s][%(levelname)s] - %(message)s")
    load_dotenv(os.path.join(ROOT, ".env"))
    cfg = C.compose(os.path.join(ROOT, "configs"), "eval.yaml", list(sys.argv[1:] if argv is None else argv))
    if cfg.get("extras", {}).get("print_config") and int(os.environ.get("RANK", "0")) == 0:
        import yaml

        log.info("config:\n" + yaml.safe_dump(C.to_plain(cfg), sort_keys=False, default_flow_style=False))
    return evaluate(cfg)


if __name__ == "__main__":
    main()
