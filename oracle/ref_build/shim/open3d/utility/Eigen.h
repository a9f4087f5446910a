#pragma once
#include "../../Eigen/mini_eigen.hpp"
