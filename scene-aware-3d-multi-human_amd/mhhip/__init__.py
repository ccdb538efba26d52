"""Host side of the MI355X-native SMPL optimisation path: ctypes binding of the C ABI
(``include/mhmocap_hip.h``), the build helper and the seeded synthetic model/sequences."""
