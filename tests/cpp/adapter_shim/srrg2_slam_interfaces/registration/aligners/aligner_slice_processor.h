// test shim (tests/cpp/adapter_shim/srrg_shim.h): stands in for <srrg2_slam_interfaces/registration/aligners/aligner_slice_processor.h> when compile-checking adapters/srrg/
#pragma once
#include "srrg_shim.h"
