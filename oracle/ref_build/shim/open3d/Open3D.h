// NOT Open3D (see geometry/PointCloud.h).  The real umbrella header drags in most of the standard library; helpers.cpp relies on that.
#pragma once
#include <algorithm>
#include <iostream>
#include <numeric>
#include <string>
#include <unordered_map>
#include <unordered_set>

#include "geometry/KDTreeFlann.h"
#include "geometry/PointCloud.h"
