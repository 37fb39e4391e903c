// exploration_cart: non-ROS entry point with the differential-drive body-twist model
#include "exploration_main.hpp"

int main(int argc, char** argv) { return exploration_main<ee::models::SimpleCart>(argc, argv, true); }
