// NOT Open3D (see Eigen/eigen_shim.hpp)
#pragma once
#include "geometry/PointCloud.h"
#include "io/PointCloudIO.h"
#include "pipelines/registration/Feature.h"
#include "pipelines/registration/GeneralizedICP.h"
#include "pipelines/registration/PoseGraph.h"
#include "pipelines/registration/Registration.h"
#include "utility/Eigen.h"
#include "utility/Helper.h"
