#pragma once
#include "PointCloud.h"
