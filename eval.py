"""Entry point with the reference's CLI:  python eval.py task_name=inference [ckpt_path=…] [target_dir=null]
(reference: src/eval.py:102-173).  Composes configs/eval.yaml (hydra if installed, otherwise the built-in
composer), instantiates datamodule / model / trainer from their `_target_`s, loads the checkpoint with the
reference's key contract and runs ``trainer.predict``; when ``target_dir`` holds reference ensembles the samples are then
scored (src/eval.py:47-99) with the device metrics of str2str_amd/metrics (validity, bonding validity, JS-PwD, JS-TICA, JS-Rg)
into the reference's tab-separated ``metrics_<tag>_<mmdd-HH-MM>.csv``."""
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


def evaluate_prediction(pred_dir: str, target_dir: str = None, tag: str = None):
    """reference src/eval.py:47-99: one row per target, one column per metric, plus the mean row."""
    from time import strftime

    import numpy as np
    import pandas as pd

    from str2str_amd.common.pdb_utils import extract_backbone_coords
    from str2str_amd.metrics import metrics

    if target_dir is None or not os.path.isdir(target_dir):
        log.warning(f"target_dir {target_dir} does not exist. Skip evaluation.")
        return {}
    assert os.path.isdir(pred_dir), f"pred_dir {pred_dir} is not a directory."
    targets = [d.replace(".pdb", "") for d in os.listdir(target_dir)]
    output_dir = os.path.dirname(os.path.dirname(os.path.abspath(pred_dir)))
    tag = tag if tag is not None else "dev"
    fns = {"val_clash": metrics.validity, "val_bond": metrics.bonding_validity, "js_pwd": metrics.js_pwd, "js_rg": metrics.js_rg,
           "js_tica": metrics.js_tica}   # the reference's five columns, in its order (src/eval.py:64-70)
    eval_res = {k: {} for k in fns}
    for target in targets:
        pred_file = os.path.join(pred_dir, f"{target}.pdb")
        if not os.path.isfile(pred_file):
            continue
        ca = {"target": extract_backbone_coords(os.path.join(target_dir, f"{target}.pdb")), "pred": extract_backbone_coords(pred_file)}
        for name, fn in fns.items():
            try:
                res = fn(ca, ref_key="target") if name.startswith("js_") else fn(ca)
            except (ValueError, NotImplementedError) as e:   # e.g. fewer reference frames than the TICA lag time: the other columns stand
                log.warning(f"{name} on {target}: {e}")
                eval_res[name][target] = float("nan")
                continue
            eval_res[name][target] = res[0]["pred"] if name == "js_tica" else res["pred"]
    df = pd.DataFrame.from_dict(eval_res)
    df.loc["mean"] = np.around(df.mean(), decimals=4)
    df.to_csv(os.path.join(output_dir, f"metrics_{tag}_{strftime('%m%d-%H-%M')}.csv"), index=True, sep="\t")
    return df.loc["mean"]


def evaluate(cfg):
    pred_dir = cfg.get("pred_dir")
    if pred_dir and os.path.isdir(pred_dir):
        log.info(f"Found pre-computed prediction directory {pred_dir}.")
        return evaluate_prediction(pred_dir, target_dir=cfg.get("target_dir"), tag=cfg.get("task_name"))
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
    if cfg.get("seed") is not None:   # (an extra key of this build: `seed=7` fixes the host noise stream; every rank seeds alike --
        torch.manual_seed(int(cfg.get("seed")))   #  the ranks then draw each chunk's noise identically and slice it, see sampler.py)
    log.info("Starting predictions.")
    pred_dir = trainer.predict(model=model, dataloaders=dataloaders, ckpt_path=ckpt_path)[-1]
    log.info(f"Samples written under {pred_dir}.")
    if int(os.environ.get("RANK", "0")) == 0 and cfg.get("target_dir"):
        log.info(f"metrics: {dict(evaluate_prediction(pred_dir, target_dir=cfg.get('target_dir'), tag=cfg.get('task_name')))}")
    return pred_dir


def main(argv=None):
    logging.basicConfig(level=logging.INFO, format="[%(asctime)s][%(name)s][%(levelname)s] - %(message)s")
    load_dotenv(os.path.join(ROOT, ".env"))
    args = list(sys.argv[1:] if argv is None else argv)
    cfg = C.compose(os.path.join(ROOT, "configs"), "eval.yaml", args)
    # `trainer.devices=N` / `trainer=ddp` from a bare shell: start N ranks, one per GPU, as Lightning does behind the reference's
    # trainer.predict (src/eval.py:129,154; configs/trainer/ddp.yaml).  Inside a torch.distributed.run job this is one of the ranks.
    from str2str_amd.utils.launch import in_distributed_job, relaunch, resolve_devices

    n_dev = resolve_devices(cfg.trainer.get("devices", 1)) if cfg.trainer.get("accelerator", "gpu") != "cpu" else 1
    if n_dev > 1 and not in_distributed_job() and not cfg.get("dry_run") and not (cfg.get("pred_dir") and os.path.isdir(cfg.get("pred_dir"))):
        rc = relaunch(n_dev, __file__, args)
        if argv is None:
            sys.exit(rc)
        if rc:
            raise RuntimeError(f"eval.py: the {n_dev}-rank job exited with code {rc}")
        return None
    if cfg.get("extras", {}).get("print_config") and int(os.environ.get("RANK", "0")) == 0:
        import yaml

        log.info("config:\n" + yaml.safe_dump(C.to_plain(cfg), sort_keys=False, default_flow_style=False))
    return evaluate(cfg)


if __name__ == "__main__":
    main()
