"""Frame -> backbone-atom projection (host-side mirror of the reference's
``src/common/all_atom.py:141-173`` ``compute_backbone``), one HIP launch (csrc/rigid_kernels.hip).

Only the five backbone atoms are ever non-zero on the sampling path (the reference tiles psi over
all 7 torsions and fills side-chain slots of atom14 with values nobody reads; atom37 keeps them 0).
"""
from __future__ import annotations

from typing import Optional

import torch

from .. import ops


def compute_backbone(bb_rigids, psi_torsions: torch.Tensor, aatype: Optional[torch.Tensor] = None, device=None,
                     _rigids7: Optional[torch.Tensor] = None):
    """-> (atom37 [*,37,3], atom37_mask [*,37], aatype, atom14 [*,14,3]); atom14 slots 5.. are zero."""
    r7 = _rigids7 if _rigids7 is not None else bb_rigids.to_tensor_7()
    if not r7.is_cuda:
        raise ops.HipLibraryError("compute_backbone runs on the HIP device only (no CPU fallback)")
    r7 = r7.type(torch.float32).contiguous()
    lead = r7.shape[:-1]
    if aatype is None:
        aatype = torch.zeros(lead, dtype=torch.long, device=r7.device)
    aatype = aatype.to(r7.device).long().contiguous()
    psi = psi_torsions.to(r7.device).type(torch.float32).contiguous()
    atom37, bb5 = ops.frames_to_backbone(r7, psi, aatype, want_atom37=True, want_atom14=True)
    atom14 = atom37.new_zeros(tuple(lead) + (14, 3))
    atom14[..., :5, :] = bb5
    return atom37, torch.any(atom37 != 0, dim=-1), aatype, atom14
