// test shim (tests/cpp/adapter_shim/srrg_shim.h): stands in for <srrg2_laser_slam_2d/mapping/scene_clipper_point_normal_2f.h> when compile-checking adapters/srrg/
#pragma once
#include "srrg_shim.h"
