"""Host side of dropout / stochastic depth for the fine-tune step.

The reference draws its masks from torch's global generator (nn.Dropout: modeling_finetune.py:51,83-85,271,
models/transformer_layer.py:236-237,394, models/decoder.py:160; timm drop_path: modeling_finetune.py:37).  Here no mask is stored
or drawn: every dropout SITE of every STEP owns a 64-bit key, and the kernels decide keep/drop per element from a keyed counter
hash of the element's coordinates (csrc/common.h `dig_drop_keep`, include/dig_hip.h `dig_dropout_t`), the same in forward and
backward.  This module derives the keys:  key(site) = splitmix64(step_seed ^ site * 0x9E3779B97F4A7C15),
step_seed = splitmix64(seed + step counter)."""
import ctypes

M64 = (1 << 64) - 1

# ---- site codes ----------------------------------------------------------------------------------------------------------
ENC_POS = 1                                   # pos_drop (modeling_finetune.py:332)


def enc_site(layer, kind):
    """kind: 0 attn_drop, 1 proj_drop, 2 drop_path(attention branch), 3 Mlp.drop (after fc2), 4 drop_path(MLP branch)."""
    return 0x100 * (layer + 1) + kind


DEC_TGT = 0x10000                             # TFDecoder.dropout on embedding + position (decoder.py:180)


def dec_site(layer, kind):
    """kind: 0 self attn_drop, 1 self proj_drop, 2 cross attn_drop, 3 cross proj_drop, 4 mlp dropout after the activation,
    5 mlp dropout after w_2 (transformer_layer.py:271,275,399,401)."""
    return 0x10000 + 0x100 * (layer + 1) + kind


def splitmix64(x):
    x = (x + 0x9E3779B97F4A7C15) & M64
    z = x
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
    return z ^ (z >> 31)


def site_key(step_seed, site):
    k = splitmix64(step_seed ^ ((site * 0x9E3779B97F4A7C15) & M64))
    return k & 0xFFFFFFFF, k >> 32


def threshold(p):
    """floor(p * 2^32): an element is dropped when its 32-bit hash is below this."""
    return min(int(float(p) * 4294967296.0), 0xFFFFFFFF)


class DropSpec(ctypes.Structure):
    """include/dig_hip.h `dig_dropout_t`."""
    _fields_ = [("k0", ctypes.c_uint), ("k1", ctypes.c_uint), ("thr", ctypes.c_uint), ("scale", ctypes.c_float),
                ("pk0", ctypes.c_uint), ("pk1", ctypes.c_uint), ("pthr", ctypes.c_uint), ("pscale", ctypes.c_float),
                ("rows_per_sample", ctypes.c_int)]


class DropPlan:
    """Keys of one training step.  `spec(site, p)` -> DropSpec (or None when nothing is dropped)."""

    def __init__(self, seed, step):
        self.step_seed = splitmix64((int(seed) + int(step)) & M64)

    def spec(self, site, p, path_site=None, path_p=0.0, rows_per_sample=0):
        if not p and not path_p:
            return None
        s = DropSpec()
        if p:
            s.k0, s.k1 = site_key(self.step_seed, site)
            s.thr, s.scale = threshold(p), 1.0 / (1.0 - float(p))
        if path_p:
            s.pk0, s.pk1 = site_key(self.step_seed, path_site)
            s.pthr, s.pscale, s.rows_per_sample = threshold(path_p), 1.0 / (1.0 - float(path_p)), rows_per_sample
        return s
