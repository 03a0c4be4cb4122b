// test shim (tests/cpp/adapter_shim/srrg_shim.h): stands in for <srrg2_laser_slam_2d/registration/correspondence_finder_nn_2d.h> when compile-checking adapters/srrg/
#pragma once
#include "srrg_shim.h"
