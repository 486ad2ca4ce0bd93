"""Entry point with the reference's command line: `python tools/train.py <config.yml>` (reference tools/train.py:
73-81) and the same YAML schema (config/default.yml of the reference; read at :23-69).

Run from the repository root as
    python -m ssd_object_detection_amd.tools.train ssd-object-detection_amd/config/default.yml
Multi-GPU (one process per GPU, RCCL):  python -m torch.distributed.run --nproc-per-node N -m ... train <cfg>"""
import argparse
import json
import logging
import os

import yaml

logger = logging.getLogger(__name__)


def load_config(yaml_file):
    with open(yaml_file, "r") as f:
        return yaml.safe_load(f)


def _make_optimizer(section, schedule):
    from .. import optimizers
    name = section["name"].lower()
    if name == "adam":
        return optimizers.Adam(schedule, **section)
    if name == "sgd":
        return optimizers.SGD(schedule, **section)
    raise ValueError


def train(config):
    import torch
    from .. import optimizers
    from ..data_loaders import SSDDataLoader
    from ..models import SSDObjectDetectionModel

    distributed = int(os.environ.get("WORLD_SIZE", "1")) > 1
    if distributed:
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        torch.distributed.init_process_group(os.environ.get("SSD_DIST_BACKEND", "nccl"))
    data = SSDDataLoader(dataset_root=config["data"]["dataset_root"], shuffle=config["data"]["shuffle"],
                         dataset=config["data"]["dataset"],
                         mini_batch=config["data"]["mini_batch"]["num_data"] if config["data"]["mini_batch"]["enable"] else 0)
    model = SSDObjectDetectionModel(classes=config["data"]["num_classes"], log_dir=config["model"]["log_dir"],
                                    distributed=distributed)
    lr_scheduler = optimizers.ExponentialDecay(initial_learning_rate=config["model"]["train"]["lr"]["initial"],
                                               decay_steps=config["model"]["train"]["lr"]["decay_step"],
                                               decay_rate=config["model"]["train"]["lr"]["decay_rate"])
    warmup_lr_scheduler = optimizers.PolynomialDecay(initial_learning_rate=config["model"]["warmup"]["lr"]["start"],
                                                     decay_steps=config["model"]["warmup"]["step"],
                                                     end_learning_rate=config["model"]["warmup"]["lr"]["end"])
    optimizer = _make_optimizer(config["model"]["train"]["optimizer"], lr_scheduler)
    warmup_optimizer = _make_optimizer(config["model"]["warmup"]["optimizer"], warmup_lr_scheduler)

    # resume (the reference has load() but no resume path): `model.resume: <checkpoint>` in the YAML
    # restores weights, Adam moments, step counters and continues with the epoch after the saved one
    start_epoch = 0
    resume = config["model"].get("resume")
    if resume:
        extra = model.load(resume)
        optimizer.iterations = int(extra.get("iterations", model.get_engine().step_count))
        warmup_optimizer.iterations = int(extra.get("warmup_iterations", 0))
        start_epoch = int(extra.get("epoch", 0))
        logger.info("Resuming from %s at epoch %d (optimizer step %d)", resume, start_epoch, optimizer.iterations)

    os.makedirs(model.get_log_dir(), exist_ok=True)
    with open(os.path.join(model.get_log_dir(), "config.json"), "w") as f:
        json.dump(config, f, sort_keys=True, indent=4, separators=(',', ':'))

    model.train(data_loader=data,
                cfg=SSDObjectDetectionModel.TrainConfig(epoch=config["model"]["train"]["epoch"],
                                                        batch_size=config["model"]["train"]["batch_size"],
                                                        optimizer=optimizer,
                                                        warmup=config["model"]["warmup"]["enable"],
                                                        warmup_optimizer=warmup_optimizer,
                                                        warmup_step=config["model"]["warmup"]["step"],
                                                        visualization_log_interval=config["model"]["log_interval"],
                                                        split_batch=config["model"]["split_train"]["enable"],
                                                        split_batch_size=config["model"]["split_train"]["batch_size"],
                                                        start_epoch=start_epoch))
    model.save(os.path.join(model.get_log_dir(), config["model"]["save"]))
    return model


if __name__ == '__main__':
    logging.basicConfig(level=logging.INFO)
    parser = argparse.ArgumentParser(description="train ssd model")
    parser.add_argument("config", type=str, help="yaml config file")
    args = parser.parse_args()
    train(load_config(args.config))
