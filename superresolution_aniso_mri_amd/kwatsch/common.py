"""settings.yaml persistence (reference kwatsch/common.py:45-68): a flat dict dumped / loaded with PyYAML."""
import argparse
from datetime import datetime

import yaml


def load_settings(fname):
    with open(fname, "r") as fp:
        return yaml.load(fp, Loader=yaml.FullLoader)


def save_settings(args, fname):
    with open(fname, "w") as fp:
        yaml.dump(vars(args), fp)


def loadExperimentSettings(fname):
    return argparse.Namespace(**load_settings(fname))


def saveExperimentSettings(args, fname):
    with open(fname, "w") as fp:
        yaml.dump(args if isinstance(args, dict) else vars(args), fp)


def generate_exper_id(exper_id=None):
    stamp = datetime.now().strftime("%m%d%H%M")
    return stamp if exper_id is None else exper_id + "_" + stamp
