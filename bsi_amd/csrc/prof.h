#pragma once
#include <hip/hip_runtime.h>

#include "../../include/bsi_hip.h"

void bsi_prof_begin(int cls, hipStream_t s);
void bsi_prof_end(int cls, hipStream_t s);

struct ProfScope {
    int cls;
    hipStream_t s;
    ProfScope(int c, hipStream_t st) : cls(c), s(st) { bsi_prof_begin(cls, s); }
    ~ProfScope() { bsi_prof_end(cls, s); }
};
