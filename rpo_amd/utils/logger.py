"""Metrics sink with the reference's interface (rpo/utils/logger.py:5-25).

``Logger(keys, epochs, times, name)`` preallocates ``epochs * times`` float64 slots per key; ``add(**kw)`` writes one
row and raises ``StopIteration`` once full; ``save(path)`` pickles the object to ``path_<yy_mm_dd_HH_MM_SS>``.
``add_rows`` is the vectorised-trainer extension: many rows at once from arrays harvested off the GPU.

On-disk compatibility: the pickle names the class ``rpo.utils.logger.Logger`` (``__module__`` below) and the instance
carries exactly the reference's attributes (``tracker``, ``name``, ``pointer``, ``epochs``, ``times``), so a file written
here loads beside the reference's package -- and the reference's plotting code -- and vice versa.
"""
import datetime
import pickle

import numpy as np


class Logger(object):

    def __init__(self, keys, epochs=1000, times=3, name="twddpg"):
        self.epochs, self.times, self.name = epochs, times, name
        self.tracker = dict((k, np.zeros((epochs * times,))) for k in keys)
        self.pointer = 0

    @property
    def capacity(self):
        return self.epochs * self.times

    def add(self, **kwargs):
        if self.pointer >= self.capacity:
            raise StopIteration("logger is full!")
        for key, value in kwargs.items():
            self.tracker[key][self.pointer] = value
        self.pointer += 1

    def add_rows(self, **columns):
        """Append len(column) rows; rows beyond the capacity raise StopIteration like ``add`` does."""
        n = len(next(iter(columns.values())))
        if self.pointer + n > self.capacity:
            raise StopIteration("logger is full!")
        for key, values in columns.items():
            self.tracker[key][self.pointer:self.pointer + n] = values
        self.pointer += n

    def save(self, path):
        import importlib
        obj, cls = self, getattr(importlib.import_module("rpo.utils.logger"), "Logger")
        if cls is not type(self):              # running beside the reference's own package: hand it an instance of ITS class
            obj = cls.__new__(cls)
            obj.__dict__.update(self.__dict__)
        stamp = datetime.datetime.now().strftime("%y_%m_%d_%H_%M_%S")
        with open("%s_%s" % (path, stamp), "wb") as f:
            pickle.dump(obj, f)


# the class is pickled by reference: name it the way the reference's files do (rpo/utils/logger.py:23-25)
Logger.__module__ = "rpo.utils.logger"
