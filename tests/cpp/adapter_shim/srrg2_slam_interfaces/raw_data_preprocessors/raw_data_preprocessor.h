// test shim (tests/cpp/adapter_shim/srrg_shim.h): stands in for <srrg2_slam_interfaces/raw_data_preprocessors/raw_data_preprocessor.h> when compile-checking adapters/srrg/
#pragma once
#include "srrg_shim.h"
