"""grim -- HLA graph imputation, MI355X-native engine behind the py-graph-imputation API."""

__organization__ = "NMDP/CIBMTR Bioinformatics (API); MI355X engine: this repository"
__version__ = "0.1.1+mi355x.r1"
