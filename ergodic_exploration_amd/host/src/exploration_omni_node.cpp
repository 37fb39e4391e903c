// exploration_omni: non-ROS entry point with the omni-directional body-twist model
#include "exploration_main.hpp"

int main(int argc, char** argv) { return exploration_main<ee::models::Omni>(argc, argv, false); }
