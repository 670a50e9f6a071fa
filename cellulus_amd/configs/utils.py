"""Converters shared by the config classes (cf. cellulus/configs/utils.py:4-18)."""

from pathlib import Path


def to_config(cls):
    """Converter: nested dict (from the toml) -> `cls`; None and ready instances pass through."""

    def converter(value):
        if value is None or isinstance(value, cls):
            return value
        return cls(**value)

    return converter


def to_path(path):
    return None if path is None else Path(path)
