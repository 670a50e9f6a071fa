"""Utilities — ``get_logger`` as in cellulus/utils/__init__.py:6-7."""

from typing import List

from .logger import Logger


def get_logger(keys: List[str], title: str) -> Logger:
    return Logger(keys, title)
