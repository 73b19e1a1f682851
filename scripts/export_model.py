#!/usr/bin/env python
"""Pack a trained Cubegan for distribution, in the layout `TTSCube.load` and the reference's downloader
(cube/io_utils/repository.py:27-61) read — counterpart of the reference's scripts/export_model.py:

  <in>.last / .yaml / .encodings  ->  <in>.model            inference weights: the discriminators and the dummy parameter dropped
                                      <out>-00, <out>-01 …  a gzip tar of cubegan.{model,yaml,encodings} [+ phonemizer.{model,encodings}]
                                                            cut into volumes of at most 49 MiB (the hosting limit the reference's format works around)
                                      <out>.yaml            {version, phonemizer: sentence, synthesis: cubegan, language, description}
"""
import argparse
import io
import os
import sys
import tarfile

import yaml

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ttscube_amd.io_utils.io_cubegan import CubeganEncodings  # noqa: E402
from ttscube_amd.networks.cubegan import Cubegan  # noqa: E402

VOLUME_BYTES = 49 * 1024 * 1024
TRAINING_ONLY = ('_mpd', '_msd', '_dummy')


def strip_to_inference_weights(base):
    """<base>.last -> <base>.model without the training-only sub-modules (Cubegan.load is non-strict, so a train=False model loads it)"""
    with open(base + '.yaml') as f:
        conditioning = yaml.load(f, yaml.Loader)['conditioning']
    model = Cubegan(CubeganEncodings(base + '.encodings'), conditioning=conditioning, train=True)
    model.load(base + '.last')
    for name in TRAINING_ONLY:
        if hasattr(model, name):
            delattr(model, name)
    model.save(base + '.model')


def archive_members(base, phonemizer):
    members = [(base + '.' + ext, 'cubegan.' + ext) for ext in ('model', 'yaml', 'encodings')]
    if phonemizer:
        members += [(phonemizer + '.sacc.best', 'phonemizer.model'), (phonemizer + '.encodings', 'phonemizer.encodings')]
    return members


def write_volumes(blob, out_base):
    """<out_base>-NN files of at most VOLUME_BYTES each; returns how many"""
    count = 0
    for start in range(0, len(blob), VOLUME_BYTES):
        with open('%s-%02d' % (out_base, count), 'wb') as f:
            f.write(blob[start:start + VOLUME_BYTES])
        count += 1
    return count


def export_model(input_model, output_model, input_phonemizer=None, version='1.0', language='en', description=''):
    strip_to_inference_weights(input_model)
    buf = io.BytesIO()
    with tarfile.open(fileobj=buf, mode='w:gz') as tar:
        for src, arcname in archive_members(input_model, input_phonemizer):
            tar.add(src, arcname)
    nvol = write_volumes(buf.getvalue(), output_model)
    with open(output_model + '.yaml', 'w') as f:
        yaml.safe_dump({'version': version, 'phonemizer': 'sentence', 'synthesis': 'cubegan', 'language': language, 'description': description}, f)
    return nvol


if __name__ == '__main__':
    ap = argparse.ArgumentParser(description=__doc__.split('\n')[0])
    ap.add_argument('--input-model', required=True, help='checkpoint base name (<base>.last, <base>.yaml, <base>.encodings)')
    ap.add_argument('--input-phonemizer', default=None, help='phonemizer base name (<base>.sacc.best, <base>.encodings), optional')
    ap.add_argument('--output-model', required=True)
    ap.add_argument('--version', default='1.0')
    ap.add_argument('--language', default='en')
    ap.add_argument('--description', default='')
    a = ap.parse_args()
    print('wrote %d volume(s)' % export_model(a.input_model, a.output_model, a.input_phonemizer, a.version, a.language, a.description))
