// test shim (tests/cpp/adapter_shim/srrg_shim.h): stands in for <srrg2_slam_interfaces/registration/correspondence_finder.h> when compile-checking adapters/srrg/
#pragma once
#include "srrg_shim.h"
