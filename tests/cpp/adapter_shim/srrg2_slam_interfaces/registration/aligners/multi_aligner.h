// test shim (tests/cpp/adapter_shim/srrg_shim.h): stands in for <srrg2_slam_interfaces/registration/aligners/multi_aligner.h> when compile-checking adapters/srrg/
#pragma once
#include "srrg_shim.h"
