// The atmosphere model from the command line: one argument, the XML configuration file (cli/atm.cpp:6-28 of the reference).
#include <cstdlib>
#include <iostream>

#include "atom_models.hpp"

int main(int argc, char** argv) {
    if (argc != 2) {
        std::cout << "\nATOM Atmosphere Model\n\nInvalid Command Line Parameter\n\nUsage:\n\t./atm <<XML configuration file path>>\n"
                  << "\tFor example: ./atm config_atm.xml\n\n";
        return 1;
    }
    try {
        cAtmosphereModel model;
        model.LoadConfig(argv[1]);
        model.Run();
    } catch (const std::exception& exc) {
        std::cerr << exc.what() << std::endl;
        return 2;
    }
    return 0;
}
