// Activation / derivative kinds of the dense-layer epilogues, and the optional per-launch timing (profile.hip).
#pragma once
#include "common.h"
#include "../../include/snerf_hip.h"

namespace snerf {

enum GemmAct { ACT_NONE = 0, ACT_SIN = 1, ACT_RELU = 2 };
enum GemmAux { AUX_NONE = 0, AUX_RELU_MASK = 2, AUX_SINREC = 3 };

int profile_begin();
int profile_end(struct ::SnerfProfile* out);

}  // namespace snerf
