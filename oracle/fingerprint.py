"""Compact fingerprints of large tensors for the golden fixtures (test infrastructure: only tests/ and tools/gen_golden_*.py import this).

A gradient of the 13.5 M-parameter text model would be a 54 MB fixture.  Instead every tensor is stored as
    norm    L2 norm (float64 accumulation)
    sum     plain sum
    probe   dot product with a N(0,1) vector seeded by the tensor's NAME (crc32) — sensitive to every element, sign and position
    idx / samples   128 evenly strided flat indices and their values — localise a mismatch
`compare` returns the largest relative deviation of (norm, probe, samples) from a stored fingerprint."""
import zlib

import numpy as np

NSAMP = 128


def probe_vector(name, n):
    return np.random.RandomState(zlib.crc32(name.encode()) & 0x7fffffff).randn(n)


def fingerprint(a, name):
    f = np.asarray(a, dtype=np.float64).reshape(-1)
    idx = np.unique(np.linspace(0, f.size - 1, min(NSAMP, f.size)).astype(np.int64))
    return {'norm': np.float64(np.sqrt((f * f).sum())), 'sum': np.float64(f.sum()), 'probe': np.float64(f @ probe_vector(name, f.size)),
            'idx': idx, 'samples': f[idx].astype(np.float32), 'size': np.int64(f.size)}


def compare(a, name, fp):
    """-> dict(norm=relative norm error, probe=|probe error| / (norm * sqrt(n)) scaled to the norm, samples=max |sample error| / max |sample|)"""
    f = np.asarray(a, dtype=np.float64).reshape(-1)
    assert f.size == int(fp['size']), (name, f.size, int(fp['size']))
    norm = float(np.sqrt((f * f).sum()))
    ref = float(fp['norm'])
    scale = max(ref, 1e-30)
    out = {'norm': abs(norm - ref) / scale,
           # a N(0,1) probe of a tensor with L2 norm `ref` has standard deviation `ref`: errors are measured on that scale
           'probe': abs(float(f @ probe_vector(name, f.size)) - float(fp['probe'])) / scale,
           'samples': float(np.abs(f[fp['idx']] - fp['samples']).max()) / max(float(np.abs(fp['samples']).max()), scale / np.sqrt(f.size))}
    return out
