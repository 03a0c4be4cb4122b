"""MI355X-native 2D scan-matching core behind the srrg2_laser_slam_2d finder/aligner plugin surface."""
