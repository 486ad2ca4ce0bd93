"""Entry point with the reference's command line: `python tools/train.py <config.yml>` (reference tools/train.py:
73-81) and the same YAML schema (config/default.yml of the reference; keys read at :23-69).

Run from the repository root as
    python -m ssd_object_detection_amd.tools.train ssd-object-detection_amd/config/default.yml
Data parallel (one process per GPU, RCCL over xGMI):
    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 -m ssd_object_detection_amd.tools.train <cfg>
`model.train.batch_size` is then the GLOBAL batch: rank r trains on its image shard of every batch (one micro-batch
of the reference's split_batch loop per rank, models/ssd_model.py:240-256), one log directory is shared, and only rank 0
writes config.json and checkpoints."""
import argparse
import json
import logging
import os

import yaml

logger = logging.getLogger(__name__)

# TrainConfig field  <-  path into the YAML document (the reference's schema, config/default.yml:17-41)
TRAIN_CONFIG_KEYS = {
    "epoch": "model/train/epoch",
    "batch_size": "model/train/batch_size",
    "warmup": "model/warmup/enable",
    "warmup_step": "model/warmup/step",
    "visualization_log_interval": "model/log_interval",
    "split_batch": "model/split_train/enable",
    "split_batch_size": "model/split_train/batch_size",
}


def load_config(yaml_file):
    with open(yaml_file, "r") as f:
        return yaml.safe_load(f)


def cfg_get(config, path):
    node = config
    for key in path.split("/"):
        node = node[key]
    return node


def _make_optimizer(section, schedule):
    from .. import optimizers
    kinds = {"adam": optimizers.Adam, "sgd": optimizers.SGD}
    kind = kinds.get(section["name"].lower())
    if kind is None:
        raise ValueError                     # reference tools/train.py:47,53
    return kind(schedule, **section)


def _init_distributed():
    """(rank, world); joins the process group when launched under torch.distributed.run."""
    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return 0, 1
    # RCCL needs one GPU per rank ON THIS NODE; with fewer GPUs than local ranks (a one-GPU box rehearsing the multi-rank
    # path) ranks share a device and the exchange goes over gloo.  The count that matters is the node's rank count
    # (LOCAL_WORLD_SIZE, set by torch.distributed.run), not the job's: 2 nodes x 8 GPUs is 16 ranks on 8 devices each.
    n_dev = torch.cuda.device_count()
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    backend = os.environ.get("SSD_DIST_BACKEND", "nccl" if n_dev >= local_world else "gloo")
    logger.info("distributed backend %s (%d ranks, %d on this node, %d devices)", backend, world,
                                     local_world, n_dev)
    if torch.cuda.is_available():
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % n_dev)
    if not torch.distributed.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.distributed.init_process_group(backend)
    return torch.distributed.get_rank(), torch.distributed.get_world_size()


def train(config):
    from .. import optimizers
    from ..data_loaders import SSDDataLoader
    from ..models import SSDObjectDetectionModel

    rank, world = _init_distributed()
    data_cfg, model_cfg = config["data"], config["model"]
    subset = data_cfg["mini_batch"]
    data = SSDDataLoader(dataset_root=data_cfg["dataset_root"], dataset=data_cfg["dataset"], shuffle=data_cfg["shuffle"],
                         mini_batch=subset["num_data"] if subset["enable"] else 0)
    model = SSDObjectDetectionModel(classes=data_cfg["num_classes"], log_dir=model_cfg["log_dir"], distributed=world > 1)

    lr, wlr = model_cfg["train"]["lr"], model_cfg["warmup"]["lr"]
    optimizer = _make_optimizer(model_cfg["train"]["optimizer"],
                                optimizers.ExponentialDecay(lr["initial"], lr["decay_step"], lr["decay_rate"]))
    warmup_optimizer = _make_optimizer(model_cfg["warmup"]["optimizer"],
                                       optimizers.PolynomialDecay(wlr["start"], model_cfg["warmup"]["step"], wlr["end"]))

    # resume (the reference has load() but no resume path): `model.resume: <checkpoint>` in the YAML
    # restores weights, Adam moments, step counters and continues with the epoch after the saved one
    start_epoch = 0
    resume = model_cfg.get("resume")
    if resume:
        extra = model.load(resume)
        optimizer.iterations = int(extra.get("iterations", model.get_engine().step_count))
        warmup_optimizer.iterations = int(extra.get("warmup_iterations", 0))
        start_epoch = int(extra.get("epoch", 0))
        logger.info("Resuming from %s at epoch %d (optimizer step %d)", resume, start_epoch, optimizer.iterations)

    if rank == 0:                            # the run's configuration next to its logs (reference tools/train.py:55-56)
        os.makedirs(model.get_log_dir(), exist_ok=True)
        text = json.dumps(config, sort_keys=True, indent=4, separators=(",", ":"))
        with open(os.path.join(model.get_log_dir(), "config.json"), "w") as f:
            f.write(text)

    fields = {name: cfg_get(config, path) for name, path in TRAIN_CONFIG_KEYS.items()}
    fields.update(optimizer=optimizer, warmup_optimizer=warmup_optimizer, start_epoch=start_epoch)
    model.train(data_loader=data, cfg=SSDObjectDetectionModel.TrainConfig(**fields))
    model.save(os.path.join(model.get_log_dir(), model_cfg["save"]))      # rank 0 writes; the others wait
    return model


if __name__ == '__main__':
    logging.basicConfig(level=logging.INFO)
    parser = argparse.ArgumentParser(description="train ssd model")
    parser.add_argument("config", type=str, help="yaml config file")
    args = parser.parse_args()
    train(load_config(args.config))
