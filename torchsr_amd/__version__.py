# mirrors torchsr/__version__.py:13
VERSION = '0.1.0'
