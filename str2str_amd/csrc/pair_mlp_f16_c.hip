// Translation unit 3 of pair_mlp_f16.hip: the edge embedding; see S2S_PM_PART there.
#define S2S_PM_PART 3
#include "pair_mlp_f16.hip"
