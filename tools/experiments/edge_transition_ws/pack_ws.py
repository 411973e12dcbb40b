"""Host side of the width-split edge-transition experiment: the per-wave weight stream (see README.md here)."""
import torch

from str2str_amd.ops import pack_f16x2_layer


def pack_f16x3_stream_ws(w1_edge: torch.Tensor, w2: torch.Tensor, wf: torch.Tensor) -> torch.Tensor:
    """The weight stream of the WIDTH-SPLIT edge transition (csrc/edge_transition_ws.hip) as int16: the same 960 fragments of 1 KiB as
    ``pack_f16x3_stream``, ordered per WAVE w (hidden tiles 3 w .. 3 w + 2, output tile w): 4 x 240 KiB,
      L1 [round r 3][k-step 8][plane 2]                  layer-1 output tile 3 w + r over the 8 k-steps of the edge row
      L2 [round r 3][k-step in round 8][tile 3][plane 2] layer-2 k-step 2 (3 v + r) + u (the round's producer wave v, half u: kk = 2 v + u)
                                                         into the wave's three output tiles
      LF [round r 3][k-step in round 8][plane 2]         final layer, same k-steps, output tile w."""
    l1, l2, lf = pack_f16x2_layer(w1_edge), pack_f16x2_layer(w2), pack_f16x2_layer(wf)   # [k-step][tile][plane][64][8]
    rounds = [[2 * (3 * v + r) + u for v in range(4) for u in range(2)] for r in range(3)]   # k-steps of a round, in kk order
    per_wave = []
    for w in range(4):
        p1 = torch.stack([l1[:, 3 * w + r] for r in range(3)])                                    # [r, ks, plane, 64, 8]
        p2 = torch.stack([l2[rounds[r]][:, 3 * w:3 * w + 3] for r in range(3)])                   # [r, kk, t, plane, 64, 8]
        pf = torch.stack([lf[rounds[r]][:, w] for r in range(3)])                                 # [r, kk, plane, 64, 8]
        per_wave.append(torch.cat([p1.reshape(-1), p2.reshape(-1), pf.reshape(-1)]))
        assert per_wave[-1].numel() * 2 == 240 * 1024
    return torch.cat(per_wave).contiguous().view(torch.int16)
