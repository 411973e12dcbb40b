// Translation unit 2 of pair_mlp_f16.hip: the edge transition for chains below 32 residues (per-lane row seeds); see S2S_PM_PART there.
#define S2S_PM_PART 2
#include "pair_mlp_f16.hip"
