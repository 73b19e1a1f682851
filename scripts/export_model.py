#!/usr/bin/env python
"""Counterpart of the reference's scripts/export_model.py:12-65: drop _mpd/_msd/_dummy, save <in>.model, tar
cubegan.{model,yaml,encodings} (+ phonemizer.{model,encodings} when given), split into 49 MiB volumes <out>-NN, write the
<out>.yaml descriptor that `TTSCube.load` / cube/io_utils/repository.py expect."""
import optparse
import os
import sys
import tarfile

import yaml

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ttscube_amd.io_utils.io_cubegan import CubeganEncodings  # noqa: E402
from ttscube_amd.networks.cubegan import Cubegan  # noqa: E402


def export_model(input_model, output_model, input_phonemizer=None, version='1.0', language='en', description=''):
    enc = CubeganEncodings('{0}.encodings'.format(input_model))
    conf = yaml.load(open('{0}.yaml'.format(input_model)), yaml.Loader)
    model = Cubegan(enc, conditioning=conf['conditioning'], train=True)
    model.load('{0}.last'.format(input_model))
    del model._mpd
    del model._msd
    if hasattr(model, '_dummy'):
        del model._dummy
    model.save('{0}.model'.format(input_model))
    tar = tarfile.open('{0}.tar.gz'.format(output_model), 'w:gz')
    for ext in ['model', 'yaml', 'encodings']:
        tar.add('{0}.{1}'.format(input_model, ext), 'cubegan.{0}'.format(ext))
    if input_phonemizer:
        for src, dst in zip(['sacc.best', 'encodings'], ['model', 'encodings']):
            tar.add('{0}.{1}'.format(input_phonemizer, src), 'phonemizer.{0}'.format(dst))
    tar.close()
    CHUNK = 49 * 1024 * 1024
    counter = 0
    with open('{0}.tar.gz'.format(output_model), 'rb') as f_in:
        while True:
            chunk = f_in.read(CHUNK)
            if not chunk:
                break
            with open('{0}-{1:02d}'.format(output_model, counter), 'wb') as f_out:
                f_out.write(chunk)
            counter += 1
    os.unlink('{0}.tar.gz'.format(output_model))
    yaml.safe_dump({'version': version, 'phonemizer': 'sentence', 'synthesis': 'cubegan', 'language': language,
                    'description': description}, open('{0}.yaml'.format(output_model), 'w'))
    return counter


if __name__ == '__main__':
    parser = optparse.OptionParser()
    parser.add_option('--input-model', dest='input_model')
    parser.add_option('--input-phonemizer', dest='input_phonemizer')
    parser.add_option('--output-model', dest='output_model')
    parser.add_option('--version', dest='version', default='1.0')
    parser.add_option('--language', dest='language', default='en')
    parser.add_option('--description', dest='description', default='')
    (params, _) = parser.parse_args(sys.argv)
    n = export_model(params.input_model, params.output_model, params.input_phonemizer, params.version, params.language, params.description)
    sys.stdout.write('wrote %d volume(s)\n' % n)
