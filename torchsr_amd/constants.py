"""Defaults of the CLI; values mirror torchsr/constants.py:13-19."""
BATCH_SIZE = 64
EPOCHS = 1000
PRE_EPOCHS = 1000
TRAIN_DIR = 'dataset'
MODEL = 'ESRGAN'
