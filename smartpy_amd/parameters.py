"""SMART parameter names, default sampling ranges and value holders (counterpart of smartpy/parameters.py).

The order of `names` is the column order of every parameter matrix handed to the engine (smart.py:204,
parameters.py:25) and the default `ranges` are the bounds of the Latin hypercube (parameters.py:27-38,
lhs.py:140-143): both are observable behaviour, so the numbers below are the reference's.
"""
import csv

# name, lower and upper bound of the default sampling range, what it is (docs/_doc_src/model_description.rst)
_DEFAULTS = (
    ('T', 0.9, 1.1),            # rainfall aerial correction coefficient
    ('C', 0.0, 1.0),            # evaporation decay parameter
    ('H', 0.0, 0.3),            # quick runoff coefficient
    ('D', 0.0, 1.0),            # drain flow parameter: fraction of saturation excess diverted to drain flow
    ('S', 0.0, 0.013),          # soil outflow coefficient
    ('Z', 15.0, 150.0),         # effective soil depth [mm]
    ('SK', 1.0, 240.0),         # surface routing parameter [hours]
    ('FK', 48.0, 1440.0),       # inter flow routing parameter [hours]
    ('GK', 1200.0, 4800.0),     # groundwater routing parameter [hours]
    ('RK', 1.0, 96.0),          # river channel routing parameter [hours]
)


class Parameters(object):
    """`names` (list), `ranges` (dict name -> (lo, hi), free for the user to narrow before sampling) and `values`
    (dict name -> value, filled by one of the two setters)."""

    def __init__(self):
        self.names = [name for name, _, _ in _DEFAULTS]
        self.ranges = {name: (lo, hi) for name, lo, hi in _DEFAULTS}
        self.values = dict()

    def _assign(self, source, missing):
        """values[name] = source[name] for the ten names, in order; `missing(name)` words the complaint."""
        for name in self.names:
            if name not in source:
                raise Exception(missing(name))
            self.values[name] = source[name]

    def set_parameters_with_file(self, file_location):
        """From a `PAR_NAME,PAR_VALUE` CSV file (parameters.py:42-74); rows of other names are ignored."""
        try:
            with open(file_location, 'r', encoding='utf8') as f:
                pairs = [(row['PAR_NAME'], row['PAR_VALUE']) for row in csv.DictReader(f)]
            found = {name: float(value) for name, value in pairs if name in self.names}
        except KeyError:
            raise Exception("There is 'PAR_NAME' or 'PAR_VALUE' column in {}.".format(file_location))
        except ValueError:
            raise Exception("There is at least one incorrect parameter value in {}.".format(file_location))
        except IOError:
            raise Exception("There is no parameters file at {}.".format(file_location))
        self._assign(found, lambda name: "The parameter {} is not available in the "
                                         "parameters file at {}.".format(name, file_location))

    def set_parameters_with_dict(self, dictionary):
        """From a dict holding (at least) the ten names (parameters.py:76-104)."""
        self._assign(dictionary,
                     lambda name: "The parameter {} is not available in the dictionary provided.".format(name))
