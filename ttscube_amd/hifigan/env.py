"""``hifigan.env.AttrDict`` as used at cube/networks/cubegan.py:42 and cube/io_utils/runtime.py:49."""


class AttrDict(dict):
    def __init__(self, *args, **kwargs):
        super(AttrDict, self).__init__(*args, **kwargs)
        self.__dict__ = self
