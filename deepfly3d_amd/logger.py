"""Logging shim with the reference's logger name so `-v/-vv` behave alike (reference df3d/logger.py)."""
import logging

_NAME = "df3d.logger"


def getLogger():
    return logging.getLogger(_NAME)


def error(msg, *a, **k):
    getLogger().error(msg, *a, **k)


def warning(msg, *a, **k):
    getLogger().warning(msg, *a, **k)


def info(msg, *a, **k):
    getLogger().info(msg, *a, **k)


def debug(msg, *a, **k):
    getLogger().debug(msg, *a, **k)
