// lsim_learn.hip -- the rollout-storage, learner and fused-policy kernels of liblsim.so (include/lsim.h from lsim_rollout_act on).
// A translation unit of its own so that the simulator's code-generation flags (fast division / square root, flushed denormals: chosen for
// kernel A's instruction count, build.py) do NOT apply here: Adam's sqrt and division, the normalisations and the loss heads are compiled
// IEEE-rounded with denormals kept, like the torch kernels they replace.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/lsim.h"
#include "ls_rollout.h"
#include "ls_learn.h"
#include "ls_gemm.h"
#include "ls_policy.h"
#include "ls_amp.h"
