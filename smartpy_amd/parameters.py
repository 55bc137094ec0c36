"""SMART parameter names, default sampling ranges and value holders (mirror of smartpy/parameters.py).

The order of `names` is the column order of every parameter matrix handed to the engine
(smart.py:204, parameters.py:25) and the default `ranges` are the bounds of the Latin hypercube
(parameters.py:27-38, lhs.py:140-143); both are observable behaviour and are kept verbatim.
"""
from csv import DictReader


class Parameters(object):
    def __init__(self):
        self.names = ['T', 'C', 'H', 'D', 'S', 'Z', 'SK', 'FK', 'GK', 'RK']
        self.ranges = {
            'T': (0.9, 1.1),
            'C': (0.0, 1.0),
            'H': (0.0, 0.3),
            'D': (0.0, 1.0),
            'S': (0.0, 0.013),
            'Z': (15.0, 150.0),
            'SK': (1.0, 240.0),
            'FK': (48.0, 1440.0),
            'GK': (1200.0, 4800.0),
            'RK': (1.0, 96.0)
        }
        self.values = dict()

    def set_parameters_with_file(self, file_location):
        """PAR_NAME,PAR_VALUE CSV (parameters.py:42-74); same error messages as the reference."""
        found = dict()
        try:
            with open(file_location, 'r', encoding='utf8') as my_file:
                for row in DictReader(my_file):
                    if row['PAR_NAME'] in self.names:
                        found[row['PAR_NAME']] = float(row['PAR_VALUE'])
        except KeyError:
            raise Exception("There is 'PAR_NAME' or 'PAR_VALUE' column in {}.".format(file_location))
        except ValueError:
            raise Exception("There is at least one incorrect parameter value in {}.".format(file_location))
        except IOError:
            raise Exception("There is no parameters file at {}.".format(file_location))
        for name in self.names:
            if name not in found:
                raise Exception("The parameter {} is not available in the "
                                "parameters file at {}.".format(name, file_location))
            self.values[name] = found[name]

    def set_parameters_with_dict(self, dictionary):
        """parameters.py:76-104."""
        for name in self.names:
            try:
                self.values[name] = dictionary[name]
            except KeyError:
                raise Exception("The parameter {} is not available in the dictionary provided.".format(name))
