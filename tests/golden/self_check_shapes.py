"""State-dict recipes shared by the tests: rebuild the seeded weights that
make_golden.py loaded into the reference modules (same key order, same RNG
stream), without needing the reference."""
import torch

from oracle import ref_cpu as R


def _fill(shapes, seed, scale=0.05):
    g = torch.Generator().manual_seed(seed)
    return {k: scale * torch.randn(shp, generator=g) for k, shp in shapes}


def dec_state(num_ch_enc, seed):
    shapes = []
    for idx, name, cin, cout in R.depth_decoder_layout(num_ch_enc):
        pre = "decoder.%d.conv.conv." % idx if name[0] == "upconv" else "decoder.%d.conv." % idx
        shapes.append((pre + "weight", (cout, cin, 3, 3)))
        shapes.append((pre + "bias", (cout,)))
    return _fill(shapes, seed)


def pose_state(num_ch_enc, seed, num_input_features=1, nf=2):
    shapes = [("net.0.weight", (256, int(num_ch_enc[-1]), 1, 1)), ("net.0.bias", (256,)),
              ("net.1.weight", (256, 256 * num_input_features, 3, 3)), ("net.1.bias", (256,)),
              ("net.2.weight", (256, 256, 3, 3)), ("net.2.bias", (256,)),
              ("net.3.weight", (6 * nf, 256, 1, 1)), ("net.3.bias", (6 * nf,))]
    return _fill(shapes, seed)
