// Host-side helpers shared by the C-ABI translation units: error reporting and the global zero page.
#pragma once
#include <hip/hip_runtime.h>

int y5_fail(int code, const char* msg);           // records msg for y5_last_error(), returns code
int y5_check_launch(const char* what);            // hipGetLastError() -> status
const void* y5_zero_page();                       // device buffer of zeros (per device), nullptr on failure
int y5_num_cu();                                  // CUs persistent grids are sized for: the device's count, capped by y5_set_cu_budget()
