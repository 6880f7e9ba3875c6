"""Minimal Logger with the reference's interface (utils/logging.py:6-49): ``setup_tb`` +
``log_stat(key, value, t)``.  tensorboardX is optional; stats are always kept in memory."""
from collections import defaultdict


class Logger:
    def __init__(self, console_logger=None):
        self.console_logger = console_logger
        self.use_tb = False
        self.stats = defaultdict(list)

    def setup_tb(self, directory_name):
        try:
            from tensorboardX import SummaryWriter     # not installed in this image
            self.writer = SummaryWriter(directory_name)
            self.use_tb = True
        except Exception:
            self.use_tb = False

    def log_stat(self, key, value, t, to_sacred=True):
        self.stats[key].append((t, value))
        if self.use_tb:
            self.writer.add_scalar(key, value, t)


def get_logger():
    """console logger of the reference (utils/logging.py:34-44): root logger, one stream handler, DEBUG"""
    import logging
    logger = logging.getLogger()
    logger.handlers = []
    ch = logging.StreamHandler()
    ch.setFormatter(logging.Formatter('[%(levelname)s %(asctime)s] %(name)s %(message)s', '%H:%M:%S'))
    logger.addHandler(ch)
    logger.setLevel('DEBUG')
    return logger
