"""Stand-in for the reference's un-vendored ``hifigan`` submodule (tiberiu44/hifi-gan), same import names:
``from hifigan.models import Generator`` / ``from hifigan.env import AttrDict`` (cube/networks/cubegan.py:18-21)."""
