"""Console entry points ``train <toml>`` / ``infer <toml>`` (cellulus/cli.py:10-27)."""

import click
import tomli

from .configs import ExperimentConfig


def _load(config_file):
    print(f"Reading config from {config_file}")
    with open(config_file, "rb") as f:
        return tomli.load(f)


@click.command()
@click.argument("config_file", type=click.Path(exists=True))
def train(config_file):
    from .train import train as train_experiment

    train_experiment(ExperimentConfig(**_load(config_file)))


@click.command()
@click.argument("config_file", type=click.Path(exists=True))
def infer(config_file):
    from .infer import infer as infer_experiment

    infer_experiment(ExperimentConfig(**_load(config_file)))
