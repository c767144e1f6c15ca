import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sound_event_detection_transformer_amd import ops, lib as L
from tools.dev.time_stem import timeit  # noqa
