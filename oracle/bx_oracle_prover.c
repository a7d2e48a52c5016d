/* placeholder translation unit: the oracle's segment-skeleton prover lands here */
#include "bx_oracle.h"
