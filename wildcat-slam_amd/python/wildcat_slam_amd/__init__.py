"""MI355X-native hot path of Wildcat-SLAM's sliding-window odometry: host-side Python mirror over the C-ABI
(include/wildcat_hip.h).  The compute lives in csrc/ (HIP, gfx950); this package only binds it."""
