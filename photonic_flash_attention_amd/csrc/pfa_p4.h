// pfa_p4.h -- host interface of the persistent 4-wave forward (pfa_p4.hip; kernel: gen_fa3_fwd_p4.py)
#pragma once
#include "pfa_hip.h"

namespace pfa {
bool p4_eligible(const pfa_fa3_args* a);                            // shape / layout the assembly kernel takes
int p4_flavour(const pfa_fa3_args* a);                              // 0 plain, 1 key mask (*_km_*), 2 ragged (*_kl_*)
int p4_prepare(int device_id, int* hip_err);                        // load the code object on the device now (idempotent)
int p4_workgroups(const pfa_fa3_args* a);                           // its grid on a->device_id (0: code object not loadable there)
int p4_launch(const pfa_fa3_args* a, void* stream, int* hip_err);   // PFA_OK or PFA_ERR_*
}  // namespace pfa
