// The hydrosphere model from the command line: one argument, the XML configuration file (cli/hyd.cpp:6-28 of the reference).
#include <cstdlib>
#include <iostream>

#include "atom_models.hpp"

int main(int argc, char** argv) {
    if (argc != 2) {
        std::cout << "\nATOM Hydrosphere Model\n\nInvalid Command Line Parameter\n\nUsage:\n\t./hyd <<XML configuration file path>>\n"
                  << "\tFor example: ./hyd config_hyd.xml\n\n";
        return 1;
    }
    try {
        cHydrosphereModel model;
        model.LoadConfig(argv[1]);
        model.Run();
    } catch (const std::exception& exc) {
        std::cerr << exc.what() << std::endl;
        return 2;
    }
    return 0;
}
