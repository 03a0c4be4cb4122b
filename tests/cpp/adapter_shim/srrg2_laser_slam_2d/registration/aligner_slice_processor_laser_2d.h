// test shim (tests/cpp/adapter_shim/srrg_shim.h): stands in for <srrg2_laser_slam_2d/registration/aligner_slice_processor_laser_2d.h> when compile-checking adapters/srrg/
#pragma once
#include "srrg_shim.h"
