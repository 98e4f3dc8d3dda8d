# claim-size / scratch-stride sweep of the compacting launch (C2 kernel ms); variants u16 = -DRZ_CLAIM_UNITS_N=16
for pad in 0 64 1024 1088 10240 10304; do RZ_CLAIM_STRIDE_PAD=$pad timeout -k 10 120 python3 profiles/scripts/config_ms.py c2 2>&1 | grep -v amdgpu.ids | sed "s/^/u8 claim8 pad$pad /"; done
for pad in 0 64 1024 1088; do RAYZEN_HIP_SO=$PWD/rayzen_amd/lib/librayzen_hip_u16.so RZ_GROUPS_PER_CLAIM=10 RZ_CLAIM_STRIDE_PAD=$pad timeout -k 10 120 python3 profiles/scripts/config_ms.py c2 2>&1 | grep -v amdgpu.ids | sed "s/^/u16 claim10 pad$pad /"; done
